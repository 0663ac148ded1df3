"""Landmark front-end of the LAFS step: the 20 augmented views of a batch -> the 2 global + n local landmark mosaics the
student/teacher consume (reference lafs_train.py:535-567 driving face_landmark_4simmin_glo_loc.forward,
face_pre_pro/ViT_face.py:1316-1409).

The reference calls the frozen landmark CNN three times (B, B and 8B images), post-processes theta with a dozen small
torch ops per call, and gathers the patches with 196 + 196 + 36 sequential grid_sample launches.  Here, per step:
    1 CNN pass over all (2+n_local)*B clean views (stock PyTorch-ROCm / MIOpen, channels-last, no_grad, eval)
    2 launches of lafs_landmark_theta   (min-max to [0,111] px, + 5 px jitter, random 36-of-196 choice for the locals)
    2 launches of lafs_patch_gather_fwd (2B x 196 patches from the augmented globals, n_local*B x 36 from the locals)
all on a private HIP stream into staging buffers, so the front-end of step i+1 overlaps the training step i; `commit`
orders the hand-over into the engine's (graph-captured) input buffers on the compute stream.
"""
import torch

from . import _lib
from .ops import _p, call

f32 = torch.float32


class LandmarkFrontEnd:
    def __init__(self, landmarkcnn, batch_size, n_local=8, image_size=112, n_local_landmarks=36, jitter_px=5.0, device=None,
                 cnn_impl="hip", cnn_dtype=torch.float32):
        """cnn_impl: "hip" = the frozen CNN compiled to the HIP launch plan of landmark_cnn.HipLandmarkCNN (NHWC bf16
        activations, folded BatchNorm, 1x1 convolutions on the MFMA GEMM); "torch" = the nn.Module on PyTorch-ROCm/MIOpen in
        `cnn_dtype` (float32 reproduces the reference's CNN bit-for-bit-ish and is what the parity tests use)."""
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.cnn = landmarkcnn.to(self.device).eval()
        for p in self.cnn.parameters():                       # frozen (lafs_train.py:269: landmarkcnn.eval())
            p.requires_grad_(False)
        self.cnn_dtype, self.cnn_impl = cnn_dtype, cnn_impl
        if cnn_impl == "hip":
            from .landmark_cnn import HipLandmarkCNN
            self.hipcnn = HipLandmarkCNN(self.cnn, self.device)
        elif cnn_impl == "torch":
            self.cnn.stn.to(memory_format=torch.channels_last)
            if cnn_dtype != torch.float32:
                self.cnn.stn.to(cnn_dtype)
        else:
            raise ValueError("cnn_impl must be 'hip' or 'torch'")
        self.B, self.n_local, self.S = batch_size, n_local, image_size
        self.n_full = landmarkcnn.row_num * landmarkcnn.row_num
        self.n_loc_lm, self.jitter = n_local_landmarks, float(jitter_px)
        dev, B = self.device, batch_size
        rg, rl = int(self.n_full ** 0.5), int(n_local_landmarks ** 0.5)
        self.theta_g = torch.empty(2 * B, self.n_full, 2, device=dev, dtype=f32)
        self.theta_l = torch.empty(max(n_local, 1) * B, n_local_landmarks, 2, device=dev, dtype=f32)
        self.stage_g = torch.empty(2 * B, 3, 8 * rg, 8 * rg, device=dev, dtype=f32)
        self.stage_l = torch.empty(max(n_local, 1) * B, 3, 8 * rl, 8 * rl, device=dev, dtype=f32)
        self.stream = torch.cuda.Stream(device=dev)
        self.ready = torch.cuda.Event()
        self.consumed = torch.cuda.Event()
        self.consumed.record()
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(torch.initial_seed() & 0x7FFFFFFF)

    @torch.no_grad()
    def _raw_landmarks(self, clean):
        """[N,3,S,S] -> raw regressor output [N, 2*n_full] (ViT_face.py:1338-1344: trunk, mean pool, Dropout(eval)+Linear)."""
        if self.cnn_impl == "hip":
            return self.hipcnn(clean.float())
        x = clean.to(self.cnn_dtype).contiguous(memory_format=torch.channels_last)
        return self.cnn.output_layer(self.cnn.stn(x).float().mean(dim=(-2, -1))).float().contiguous()

    @torch.no_grad()
    def prefetch(self, views, noise=None, sel=None, produced=None):
        """views: list of 2*(2+n_local) tensors [B,3,S,S] in the reference's order (clean, augmented pairs: g0, g0', g1, g1',
        l0, l0', ...) or one stacked tensor [2*(2+n_local), B, 3, S, S].  noise [(2+n_local)*B, n_full, 2] ~ N(0,1) and
        sel int32 [n_local*B, 36] may be supplied (tests); otherwise they are drawn on the device.  `produced`: event recorded
        when the views became valid; without it the front-end stream waits for everything enqueued on the current stream so
        far (which would serialise it behind a training step launched in between)."""
        B, nl, dev = self.B, self.n_local, self.device
        cur = torch.cuda.current_stream(dev)
        if produced is not None:
            self.stream.wait_event(produced)
        else:
            self.stream.wait_stream(cur)                       # the views were produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.consumed)              # staging buffers free again
            for t_ in ([views] if torch.is_tensor(views) else views):
                if t_.is_cuda:
                    t_.record_stream(self.stream)             # allocated on the caller's stream, consumed here
            if torch.is_tensor(views):
                v = views.to(dev, f32)
                clean, aug = v[0::2].reshape(-1, 3, self.S, self.S), v[1::2].reshape(-1, 3, self.S, self.S)
            else:
                clean = torch.cat([t.to(dev, f32) for t in views[0::2]])
                aug = torch.cat([t.to(dev, f32) for t in views[1::2]])
            if clean.shape[0] != (2 + nl) * B:
                raise _lib.LafsHipError(f"expected {(2 + nl) * B} clean views, got {clean.shape[0]}")
            aug = aug.contiguous()
            t = self._raw_landmarks(clean)
            if noise is None:
                noise = torch.randn((2 + nl) * B, self.n_full, 2, device=dev, generator=self.gen)
            if sel is None and nl:
                sel = torch.randint(0, self.n_full, (nl * B, self.n_loc_lm), device=dev, generator=self.gen, dtype=torch.int32)
            noise = noise.to(dev, f32).contiguous()
            call("lafs_landmark_theta", _p(t), 2 * B, self.n_full, _p(noise), self.jitter, None, self.n_full, _p(self.theta_g))
            call("lafs_patch_gather_fwd", _p(aug), _p(self.theta_g), 2 * B, self.S, self.n_full, _p(self.stage_g))
            if nl:
                sel = sel.to(dev, torch.int32).contiguous()
                call("lafs_landmark_theta", _p(t[2 * B:]), nl * B, self.n_full, _p(noise[2 * B:]), self.jitter, _p(sel),
                     self.n_loc_lm, _p(self.theta_l))
                call("lafs_patch_gather_fwd", _p(aug[2 * B:]), _p(self.theta_l), nl * B, self.S, self.n_loc_lm, _p(self.stage_l))
            self.ready.record(self.stream)
            for x in (clean, aug, t, noise, sel):              # keep the allocator from recycling them under the side stream
                if x is not None:
                    x.record_stream(self.stream)

    def commit(self, engine):
        """Hand the staged mosaics to the engine's input buffers on the current (compute) stream."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready)
        engine.in_global_all.copy_(self.stage_g)
        if self.n_local:
            engine.in_local_all.copy_(self.stage_l)
        self.consumed.record(cur)

    def __call__(self, views, engine, **kw):
        self.prefetch(views, **kw)
        self.commit(engine)
