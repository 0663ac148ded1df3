"""Training plan of the TRAINABLE landmark CNN on the HIP kernels (second half).

Part-fViT with ``with_land=True`` -- the model train_largescale.py:432,542-561 really fine-tunes -- regresses its 196 landmarks
with a MobileNetV3-large trunk that is TRAINED with the rest of the network (reference face_pre_pro/mobilenet.py:224-313,
ViT_face.py:679-711): BatchNorm in training mode, Dropout(0.5) in front of the regressor, gradients arriving through the
patch gather.  Rounds 1-2 ran this branch on torch autograd over MIOpen (~700 launches, 10 of 34 ms per step at batch 128).
Here it is a launch plan over NHWC fp16 activations (the reference's autocast format; channels padded to 32,
csrc/landmark_train.hip; round 3 stored bf16 and sat 9 % from the fp32 reference where fp16 sits at ~1 %):

    forward   im2col stem -> [1x1 conv = lafs_gemm_nt -> BN statistics -> BN apply + activation] ... depthwise kernels, squeeze-
              excite (pool, two small GEMMs, rescale), project conv + BN + residual ... pool -> Dropout -> Linear -> min-max theta
    backward  the mirror image: BN backward (reduce + apply, activation derivative recomputed), lafs_wgrad for every 1x1 / FC
              weight, lafs_gemm_nt on W^T shadows for the input gradients, depthwise data / weight gradients, SE backward

Parameters stay fp32 in the fine-tune arena (no copies: gamma / beta / depthwise weights are read in place, their gradients
accumulated in place); the padded fp16 operand images of the 1x1 / FC weights are refreshed by one table-driven launch whenever the
arena's master weights changed, the padded weight gradients folded into the arena by one launch per backward.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib, ops
from .ops import _p, call

f32, h16 = torch.float32, torch.float16      # 16-bit storage of this plan: IEEE fp16 (round 3: bf16), see csrc/landmark_train.hip
GRAD_TARGET = 1024.0                          # max |scaled gradient| entering the CNN backward (power-of-two scale chosen on the device)


def _pad32(c):
    return (c + 31) // 32 * 32


def _act_code(m):
    if isinstance(m, nn.ReLU):
        return _lib.ACT_RELU
    if isinstance(m, nn.Hardswish):
        return _lib.ACT_HSWISH
    if isinstance(m, nn.Hardsigmoid):
        return _lib.ACT_HSIGMOID
    return _lib.ACT_NONE


def _bn_momentum(bn):
    """nn.BatchNorm2d(momentum=None) means a cumulative moving average (factor 1 / num_batches_tracked), which this plan's constant
    factor cannot express: refuse it instead of substituting a value (the reference's MobileNetV3 uses the default 0.1)."""
    if bn.momentum is None:
        raise _lib.LafsHipError("HipLandmarkTrainer: BatchNorm2d(momentum=None) (cumulative average) is not supported")
    return float(bn.momentum)


class _Tables:
    """Builder of the two device tables: padded bf16 operand images (W and W^T) and padded fp32 gradient images."""

    def __init__(self):
        self.cast, self.cast_blocks, self.cast_size = [], [0], 0          # entries of lafs_cnn_pad_cast_table
        self.fold, self.fold_blocks, self.grad_size = [], [0], 0          # entries of lafs_cnn_unpad_add_table
        self.dwl, self.dwl_blocks, self.dw_size = [], [0], 0              # entries of lafs_cnn_dw_layout_table

    def operand(self, src_off, rows, cols, prow, pcol, transpose):
        """Image [prow, pcol] of src [rows, cols] (transpose: [pcol-padded cols as rows]); returns its offset (bf16 elements)."""
        r, c = (pcol, prow) if transpose else (prow, pcol)                 # image shape (rows, ld)
        off = self.cast_size
        self.cast.append([src_off, rows, cols, off, c, 1 if transpose else 0, r, 0])
        self.cast_blocks.append(self.cast_blocks[-1] + (r * c + 1023) // 1024)
        self.cast_size += (r * c + 63) // 64 * 64
        return off

    def depthwise(self, src_off, C, kk, ld):
        """Tap-major fp32 image [kk, ld] of a depthwise weight [C, 1, k, k] and the same-shaped gradient image (folded back transposed);
        returns (weight image offset, gradient image offset)."""
        off = self.dw_size
        self.dwl.append([src_off, C, kk, off, ld, 0, 0, 0])
        self.dwl_blocks.append(self.dwl_blocks[-1] + (kk * ld + 255) // 256)
        self.dw_size += (kk * ld + 63) // 64 * 64
        goff = self.grad_size
        self.fold.append([goff, C, kk, ld, src_off, 1, 0, 0])                # grad[c][t] += image[t][c]
        self.fold_blocks.append(self.fold_blocks[-1] + (C * kk + 255) // 256)
        self.grad_size += (kk * ld + 63) // 64 * 64
        return off, goff

    def gradient(self, rows, cols, prow, pcol, grad_off):
        """Padded fp32 image [prow, pcol] whose [rows, cols] corner is added to arena.grad[grad_off ...]; returns its offset."""
        off = self.grad_size
        self.fold.append([off, rows, cols, pcol, grad_off, 0, 0, 0])
        self.fold_blocks.append(self.fold_blocks[-1] + (rows * cols + 255) // 256)
        self.grad_size += (prow * pcol + 63) // 64 * 64
        return off


class HipLandmarkTrainer:
    def __init__(self, model, arena, batch_size, image_size=112, device=None):
        """model: ViT_face_landmark_patch8(with_land=True) whose parameters live in `arena` (vision_transformer.attach_arena)."""
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.model, self.arena, self.N, self.S = model, arena, batch_size, image_size
        dev, a = self.device, arena
        feats = model.stn.features
        off = lambda name: a.offsets[name]
        T = _Tables()
        sums = [0]                                                       # running size of the BatchNorm scratch (floats)

        def bn_spec(prefix, bn, C):
            s = dict(C=C, g=off(prefix + ".weight"), b=off(prefix + ".bias"), rm=bn.running_mean, rv=bn.running_var, eps=float(bn.eps),
                     mom=_bn_momentum(bn), nbt=bn.num_batches_tracked, sums=sums[0], dsums=sums[0] + 2 * C,
                     stat=None)
            sums[0] += 4 * C
            return s

        def conv_spec(name, cout, cin, po, pi, need_t=True):
            o = off(name)
            return dict(w=T.operand(o, cout, cin, po, pi, False), wt=T.operand(o, cout, cin, po, pi, True) if need_t else None,
                        gw=T.gradient(cout, cin, po, pi, o), cout=cout, cin=cin, po=po, pi=pi)

        # stem: conv [16, 3, 3, 3] = [16, 27] on im2col rows
        self.stem = dict(conv=conv_spec("stn.features.0.0.weight", 16, 27, 32, 32, need_t=False), bn=bn_spec("stn.features.0.1", feats[0][1], 16),
                         act=_act_code(feats[0][2]))
        self.blocks = []
        cin = 16
        for i, blk in enumerate(feats[1:], start=1):
            cv = blk.conv
            cexp, cout = cv[0].out_channels, cv[7].out_channels
            k, stride = cv[3].kernel_size[0], cv[3].stride[0]
            pi, pe, po = _pad32(cin), _pad32(cexp), _pad32(cout)
            pre = f"stn.features.{i}.conv."
            L = dict(cin=cin, cexp=cexp, cout=cout, pi=pi, pe=pe, po=po, k=k, stride=stride, residual=bool(blk.residual), act=_act_code(cv[2]),
                     exp=conv_spec(pre + "0.weight", cexp, cin, pe, pi), bn1=bn_spec(pre + "1", cv[1], cexp),
                     dw=T.depthwise(off(pre + "3.weight"), cexp, k * k, pe), bn2=bn_spec(pre + "4", cv[4], cexp),
                     proj=conv_spec(pre + "7.weight", cout, cexp, po, pe), bn3=bn_spec(pre + "8", cv[8], cout), se=None)
            if not isinstance(cv[5], nn.Identity):
                h = cv[5].fc[0].out_features
                ph = _pad32(h)
                L["se"] = dict(h=h, ph=ph, fc1=conv_spec(pre + "5.fc.0.weight", h, cexp, ph, pe), fc2=conv_spec(pre + "5.fc.2.weight", cexp, h, pe, ph),
                               act1=_act_code(cv[5].fc[1]), act2=_act_code(cv[5].fc[3]))
            self.blocks.append(L)
            cin = cout
        self.c_last, self.p_last = cin, _pad32(cin)
        lin = model.output_layer[1]
        self.n_out, self.p_out = lin.out_features, _pad32(lin.out_features)
        self.drop_p = float(model.output_layer[0].p)
        self.head = conv_spec("output_layer.1.weight", self.n_out, self.c_last, self.p_out, self.p_last)
        self.head_b = off("output_layer.1.bias")
        self.head_gb = T.gradient(1, self.n_out, 1, self.p_out, self.head_b)
        # device tables and buffers
        i64 = torch.int64
        self.cast_table = torch.tensor(T.cast, dtype=i64, device=dev).view(-1)
        self.cast_starts = torch.tensor(T.cast_blocks, dtype=torch.int32, device=dev)
        self.fold_table = torch.tensor(T.fold, dtype=i64, device=dev).view(-1)
        self.fold_starts = torch.tensor(T.fold_blocks, dtype=torch.int32, device=dev)
        self.n_cast, self.cast_nblk, self.n_fold, self.fold_nblk = len(T.cast), T.cast_blocks[-1], len(T.fold), T.fold_blocks[-1]
        self.dwl_table = torch.tensor(T.dwl, dtype=i64, device=dev).view(-1)
        self.dwl_starts = torch.tensor(T.dwl_blocks, dtype=torch.int32, device=dev)
        self.n_dwl, self.dwl_nblk = len(T.dwl), T.dwl_blocks[-1]
        self.wdw = torch.zeros(T.dw_size, device=dev, dtype=f32)
        self.dw_grad_lo = min(L["dw"][1] for L in self.blocks)             # the depthwise gradient images are zeroed every backward
        self.dw_grad_hi = max(L["dw"][1] + ((L["k"] ** 2 * L["pe"] + 63) // 64 * 64) for L in self.blocks)
        self.wbf = torch.zeros(T.cast_size, device=dev, dtype=h16)
        self.gpad = torch.zeros(T.grad_size, device=dev, dtype=f32)
        self.bn_ws = torch.zeros(sums[0], device=dev, dtype=torch.float64)     # BatchNorm sums: fp64 (exact enough to be order-independent)
        # device state of the loss scaling: {scale, 1 / scale, target, found_inf, skipped backwards, clean backwards in a row, -, -}
        # (lafs_cnn_grad_scale / lafs_cnn_grad_guard: the reference's GradScaler, train_largescale.py:739, 867-880, without a host sync)
        self.gscale = torch.tensor([1.0, 1.0, GRAD_TARGET, 0.0, 0.0, 0.0, 0.0, 0.0], device=dev, dtype=f32)
        # the CNN's gradient range of the arena (stn.* and output_layer.*): what the overflow guard checks and, on inf / NaN, zeroes
        cnn = [i for i, n in enumerate(arena.names) if n.startswith(("stn.", "output_layer."))]
        assert cnn == list(range(cnn[0], cnn[-1] + 1)), "the landmark CNN's tensors must be contiguous in the arena"
        self.grad_lo = arena.offsets[arena.names[cnn[0]]]
        last = arena.names[cnn[-1]]
        self.grad_hi = arena.offsets[last] + (arena.numels[last] + _lib.CHUNK - 1) // _lib.CHUNK * _lib.CHUNK
        self._versions = None
        self.step = 0
        self.seed = 0x1A2D
        self.n_forward = 0                     # training forwards since the last flush of num_batches_tracked
        self.step_dev = None                   # device step counter of a captured step (see forward)
        self.fixed_drop = None                 # parity hook: dropout factors f32 [N, 160] (0 or 1/(1-p)) instead of the counter-based mask
        self.keep_trace, self.trace = False, []   # parity hook: clones of every block's incoming / intermediate / outgoing gradients
        self._alloc()

    # ------------------------------------------------------------------ helpers over the flat buffers
    def _w(self, offset, rows, ld):
        return self.wbf[offset: offset + rows * ld].view(rows, ld)

    def _gw(self, offset, rows, ld):
        return self.gpad[offset: offset + rows * ld].view(rows, ld)

    def _m(self, offset):                       # raw pointers into the arena
        return C.c_void_p(self.arena.master.data_ptr() + 4 * offset)

    def _g(self, offset):
        return C.c_void_p(self.arena.grad.data_ptr() + 4 * offset)

    def _ws(self, offset):
        return C.c_void_p(self.bn_ws.data_ptr() + 8 * offset)

    def _dww(self, L):                          # tap-major weight image / gradient image of a block's depthwise convolution
        return C.c_void_p(self.wdw.data_ptr() + 4 * L["dw"][0])

    def _dwg(self, L):
        return C.c_void_p(self.gpad.data_ptr() + 4 * L["dw"][1])

    def refresh_operands(self):
        """Padded bf16 images (W and W^T) of every 1x1 / FC weight from the arena's fp32 master, one launch."""
        call("lafs_cnn_pad_cast_table", _p(self.arena.master), _p(self.wbf), _p(self.cast_table), _p(self.cast_starts), self.n_cast, self.cast_nblk)
        call("lafs_cnn_dw_layout_table", _p(self.arena.master), _p(self.wdw), _p(self.dwl_table), _p(self.dwl_starts), self.n_dwl, self.dwl_nblk)

    def _alloc(self):
        N, dev = self.N, self.device
        H = self.S // 2
        mk = lambda rows, ld: torch.empty(rows, ld, device=dev, dtype=h16)
        B = dict(P=mk(N * H * H, 32), s_raw=mk(N * H * H, 32), x0=mk(N * H * H, 32), layers=[])
        wmax = 0
        wb = lambda M, n1, n2: max(int(_lib.lib().lafs_wgrad_workspace_bytes(M, n1, n2)), 0)
        wmax = max(wmax, wb(N * H * H, 32, 32))
        for L in self.blocks:
            Ho = (H + L["stride"] - 1) // L["stride"]
            R, Ro = N * H * H, N * Ho * Ho
            d = dict(H=H, Ho=Ho, R=R, Ro=Ro, e_raw=mk(R, L["pe"]), e=mk(R, L["pe"]), d_raw=mk(Ro, L["pe"]), d=mk(Ro, L["pe"]),
                     y_raw=mk(Ro, L["po"]), y=mk(Ro, L["po"]))
            if L["se"]:
                ph = L["se"]["ph"]
                d.update(zb=mk(Ro, L["pe"]), pool=mk(N, L["pe"]), hid=mk(N, ph), gate=mk(N, L["pe"]),
                         dgate=torch.empty(N, L["pe"], device=dev, dtype=f32), dg2=mk(N, L["pe"]), dhid=mk(N, ph), dpool=mk(N, L["pe"]))
                wmax = max(wmax, wb(N, L["pe"], ph), wb(N, ph, L["pe"]))
            wmax = max(wmax, wb(R, L["pe"], L["pi"]), wb(Ro, L["po"], L["pe"]))
            B["layers"].append(d)
            H = Ho
        self.H_last = H
        B["feat"] = mk(N, self.p_last)
        B["featf"] = torch.empty(N, self.p_last, device=dev, dtype=f32)
        B["featd"] = mk(N, self.p_last)
        B["t"] = torch.empty(N, self.n_out, device=dev, dtype=f32)
        B["theta"] = torch.empty(N, self.n_out // 2, 2, device=dev, dtype=f32)
        B["zero_noise"] = torch.zeros(N, self.n_out // 2, 2, device=dev, dtype=f32)
        B["dt"] = torch.empty(N, self.n_out, device=dev, dtype=f32)
        B["dt_bf"] = torch.zeros(N, self.p_out, device=dev, dtype=h16)
        B["dfeatf"] = torch.empty(N, self.p_last, device=dev, dtype=f32)
        B["dfeat"] = mk(N, self.p_last)
        # gradient ping-pong buffers, sized for the largest activation
        big = max(max(d["R"] * L["pe"], d["Ro"] * L["po"], d["R"] * L["pi"]) for d, L in zip(B["layers"], self.blocks))
        B["ga"] = torch.empty(big, device=dev, dtype=h16)
        B["gb"] = torch.empty(big, device=dev, dtype=h16)
        B["gc"] = torch.empty(big, device=dev, dtype=h16)
        wmax = max(wmax, wb(N, self.p_out, self.p_last))
        B["wg_ws"] = torch.empty(max(wmax, 16) // 4, device=dev, dtype=f32)
        self.B = B

    # ------------------------------------------------------------------ BatchNorm wrappers
    def _bn_fwd(self, spec, x, R, act, out, resid=None):
        train = self.model.training
        if train:
            call("lafs_cnn_bn_stats", _p(x), x.shape[1], R, spec["C"], self._ws(spec["sums"]))
        else:       # eval mode: the running statistics, as the sums of a batch that has exactly them; no momentum update
            call("lafs_cnn_bn_eval_sums", _p(spec["rm"]), _p(spec["rv"]), R, spec["C"], self._ws(spec["sums"]))
        if spec["stat"] is None:
            spec["stat"] = torch.empty(2 * spec["C"], device=self.device, dtype=f32)
        call("lafs_cnn_bn_apply", _p(x), x.shape[1], R, spec["C"], self._ws(spec["sums"]), self._m(spec["g"]), self._m(spec["b"]), spec["eps"],
             spec["mom"], _p(spec["rm"]) if train else None, _p(spec["rv"]) if train else None, act, _p(resid),
             resid.shape[1] if resid is not None else 0, _p(out), out.shape[1], _p(spec["stat"]))

    def _bn_bwd(self, spec, dy, x, R, act, dx, add_nc=None, HW=1):
        call("lafs_cnn_bn_bwd" if self.model.training else "lafs_cnn_bn_bwd_eval", _p(dy), dy.shape[1], _p(x), x.shape[1], R, spec["C"],
             _p(spec["stat"]), self._m(spec["g"]), self._m(spec["b"]), act, _p(add_nc), add_nc.shape[1] if add_nc is not None else 0, HW,
             self._ws(spec["dsums"]), _p(dx), dx.shape[1], self._g(spec["g"]), self._g(spec["b"]), _p(self.gscale))

    def _wgrad(self, dy, x, cs):
        """padded dW [po, pi] = dy^T x into the gradient image of conv spec `cs` (folded into the arena at the end of backward)."""
        gw = self._gw(cs["gw"], cs["po"], cs["pi"])
        ws = self.B["wg_ws"]
        call("lafs_wgrad_f16", _p(dy), dy.shape[1], _p(x), x.shape[1], _p(gw), cs["pi"], dy.shape[0], cs["po"], cs["pi"], 0, None, _p(ws),
             ws.numel() * 4)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x):
        """x f32 NCHW [N,3,S,S] (normalised, mixed batch) -> theta f32 [N, r*r, 2] in pixels; everything the backward needs is kept."""
        if not x.is_cuda or x.dtype != f32 or x.shape != (self.N, 3, self.S, self.S):
            raise _lib.LafsHipError("HipLandmarkTrainer.forward expects the float32 NCHW device batch it was built for")
        a, B, N = self.arena, self.B, self.N
        a.ensure_fresh()
        ver = tuple(p._version for p in a.params)
        if self._versions != ver or getattr(self, "_stale", True):
            self.refresh_operands()
            self._versions, self._stale = ver, False
        call("lafs_fill_zero", _p(self.bn_ws), self.bn_ws.numel() * 8)
        self.x_in = x.contiguous()
        H0 = self.S // 2
        R0 = N * H0 * H0
        call("lafs_cnn_im2col_stem", _p(self.x_in), N, self.S, _p(B["P"]))
        st = self.stem
        ops.gemm_nt(B["P"], self._w(st["conv"]["w"], 32, 32), _lib.EPI_BF16, out=B["s_raw"])
        self._bn_fwd(st["bn"], B["s_raw"], R0, st["act"], B["x0"])
        cur = B["x0"]
        for L, D in zip(self.blocks, B["layers"]):
            R, Ro, H, Ho = D["R"], D["Ro"], D["H"], D["Ho"]
            ops.gemm_nt(cur, self._w(L["exp"]["w"], L["pe"], L["pi"]), _lib.EPI_BF16, out=D["e_raw"])
            self._bn_fwd(L["bn1"], D["e_raw"], R, L["act"], D["e"])
            call("lafs_cnn_dwconv_train_fwd", _p(D["e"]), self._dww(L), N, H, H, L["pe"], L["cexp"], L["k"], L["stride"], _p(D["d_raw"]))
            se = L["se"]
            if se:
                self._bn_fwd(L["bn2"], D["d_raw"], Ro, _lib.ACT_NONE, D["zb"])
                call("lafs_cnn_pool_train", _p(D["zb"]), N, Ho * Ho, L["pe"], _p(D["pool"]), L["pe"])
                ops.gemm_nt(D["pool"], self._w(se["fc1"]["w"], se["ph"], L["pe"]), _lib.EPI_BF16_ACT, out=D["hid"], act=se["act1"])
                ops.gemm_nt(D["hid"], self._w(se["fc2"]["w"], L["pe"], se["ph"]), _lib.EPI_BF16_ACT, out=D["gate"], act=se["act2"])
                call("lafs_cnn_scale_act_out", _p(D["zb"]), _p(D["gate"]), L["pe"], N, Ho * Ho, L["pe"], L["act"], _p(D["d"]))
            else:
                self._bn_fwd(L["bn2"], D["d_raw"], Ro, L["act"], D["d"])
            ops.gemm_nt(D["d"], self._w(L["proj"]["w"], L["po"], L["pe"]), _lib.EPI_BF16, out=D["y_raw"])
            self._bn_fwd(L["bn3"], D["y_raw"], Ro, _lib.ACT_NONE, D["y"], resid=cur if L["residual"] else None)
            D["x_in"] = cur
            cur = D["y"]
        HW = self.H_last * self.H_last
        call("lafs_cnn_pool_train", _p(cur), N, HW, self.p_last, _p(B["feat"]), self.p_last)
        # Dropout(0.5) (training) + Linear(160, 2 r r): counter-based mask of (seed + 7919 step), regenerated in the backward
        call("lafs_cnn_cast_f16_f32", _p(B["feat"]), _p(B["featf"]), B["feat"].numel())
        # host counter: seed + 7919 * step per forward; with `step_dev` (a DEVICE float, e.g. hyper[HP_STEP] of a captured fine-tune
        # step) the kernels add 7919 * step themselves, so a replayed graph draws a new mask every micro-step
        self.drop_seed = self.seed if self.step_dev is not None else (self.seed + 7919 * self.step) & 0xFFFFFFFF
        self.step += 1
        self._dropout(B["featf"])
        call("lafs_cnn_cast_pad_f16", _p(B["featf"]), N, self.p_last, _p(B["featd"]), self.p_last, None)
        ops.gemm_nt(B["featd"], self._w(self.head["w"], self.p_out, self.p_last)[: self.n_out], _lib.EPI_F32,
                    bias=a.view(a.master, "output_layer.1.bias"), out=B["t"])
        n_full = self.n_out // 2
        call("lafs_landmark_theta", _p(B["t"]), N, n_full, _p(B["zero_noise"]), 0.0, None, n_full, _p(B["theta"]))
        if self.model.training and not torch.cuda.is_current_stream_capturing():
            # (a captured forward is counted per REPLAY by its owner, FinetuneEngine.micro_step: this Python code runs only at capture)
            self.n_forward += 1
        return B["theta"]

    def _dropout(self, buf):
        """Dropout(0.5) of the pooled feature (forward) / of its gradient (backward): the same counter-based mask both times."""
        if self.fixed_drop is not None:
            buf.mul_(self.fixed_drop.to(buf.device, buf.dtype))
        elif self.model.training and self.drop_p > 0:
            call("lafs_dropout_f32", _p(buf), self.p_last, self.N, self.p_last, self.drop_p, self.drop_seed, _p(self.step_dev))

    def flush_batches_tracked(self):
        """nn.BatchNorm2d.num_batches_tracked of every BatchNorm of the trunk += the training forwards run since the last flush
        (the counters only matter for state_dict parity -- momentum is a constant 0.1 -- so they are brought up to date when a
        state_dict is taken, not with a launch per step)."""
        if self.n_forward:
            for s in [self.stem["bn"]] + [L[k] for L in self.blocks for k in ("bn1", "bn2", "bn3")]:
                if s["nbt"] is not None:
                    s["nbt"].add_(self.n_forward)
            self.n_forward = 0

    # ------------------------------------------------------------------ backward
    @torch.no_grad()
    def backward(self, dtheta):
        """dtheta f32 [N, r*r, 2]: gradient of the loss w.r.t. the landmarks (from lafs_patch_gather_bwd).  Accumulates the gradients
        of stn.* and output_layer.* into the arena."""
        a, B, N = self.arena, self.B, self.N
        n_full = self.n_out // 2
        call("lafs_landmark_theta_bwd", _p(B["t"]), _p(dtheta.contiguous()), N, self.n_out, _p(B["dt"]))
        call("lafs_fill_zero", C.c_void_p(self.gpad.data_ptr() + 4 * self.dw_grad_lo), (self.dw_grad_hi - self.dw_grad_lo) * 4)
        # loss scaling (the reference's GradScaler, train_largescale.py:803-867): the gradient enters the fp16 backward multiplied by
        # the power of two that brings its largest entry to ~GRAD_TARGET, chosen on the device; it is divided out where gradients
        # leave the 16-bit domain (BatchNorm affine gradients, the fold of the padded weight gradients)
        call("lafs_cnn_grad_scale", _p(B["dt"]), B["dt"].numel(), GRAD_TARGET, _p(self.gscale), 1)
        call("lafs_cnn_cast_pad_f16", _p(B["dt"]), N, self.n_out, _p(B["dt_bf"]), self.p_out, _p(self.gscale))
        # head: dW, db, d(feature)
        hd = self.head
        gw = self._gw(hd["gw"], hd["po"], hd["pi"])
        gb = self.gpad[self.head_gb: self.head_gb + self.p_out]
        call("lafs_fill_zero", _p(gb), self.p_out * 4)
        ws = B["wg_ws"]
        call("lafs_wgrad_f16", _p(B["dt_bf"]), self.p_out, _p(B["featd"]), self.p_last, _p(gw), hd["pi"], N, hd["po"], hd["pi"], 0, _p(gb), _p(ws),
             ws.numel() * 4)
        ops.gemm_nt(B["dt_bf"], self._w(hd["wt"], hd["pi"], hd["po"]), _lib.EPI_F32, out=B["dfeatf"])
        self._dropout(B["dfeatf"])
        call("lafs_cnn_cast_pad_f16", _p(B["dfeatf"]), N, self.p_last, _p(B["dfeat"]), self.p_last, None)
        HW = self.H_last * self.H_last
        Dl, Ll = B["layers"][-1], self.blocks[-1]
        view = lambda buf, rows, ld: buf[: rows * ld].view(rows, ld)
        dy = view(B["ga"], Dl["Ro"], Ll["po"])
        call("lafs_cnn_pool_bwd", _p(B["dfeat"]), self.p_last, N, HW, self.p_last, _p(dy))
        spare = [B["gb"], B["gc"]]
        dy_buf = B["ga"]

        def take():
            return spare.pop()

        def give(buf):
            spare.append(buf)
        self.trace = []
        for L, D in zip(reversed(self.blocks), reversed(B["layers"])):
            R, Ro, H, Ho = D["R"], D["Ro"], D["H"], D["Ho"]
            tr = {"dy": dy.clone()} if self.keep_trace else None
            # y = BN3(y_raw) (+ x_in): the residual branch carries dy unchanged into the block input gradient
            b1 = take(); dy_raw = view(b1, Ro, L["po"])
            self._bn_bwd(L["bn3"], dy, D["y_raw"], Ro, _lib.ACT_NONE, dy_raw)
            self._wgrad(dy_raw, D["d"], L["proj"])
            b2 = take(); dd = view(b2, Ro, L["pe"])
            ops.gemm_nt(dy_raw, self._w(L["proj"]["wt"], L["pe"], L["po"]), _lib.EPI_BF16, out=dd)
            if tr is not None:
                tr["dd"] = dd.clone()
            se = L["se"]
            dd_raw = view(b1, Ro, L["pe"])                               # b1 (dy_raw) is dead after the two GEMMs above
            if se:
                dz = dd_raw                                              # ds * gate goes to b1, then BN2 backward b1 -> b2
                call("lafs_cnn_se_bwd", _p(dd), _p(D["zb"]), _p(D["gate"]), L["pe"], N, Ho * Ho, L["pe"], L["act"], _p(dz), _p(D["dgate"]), L["pe"])
                call("lafs_cnn_act_bwd_post", _p(D["dgate"]), None, _p(D["gate"]), D["dgate"].numel(), se["act2"], _p(D["dg2"]))
                self._wgrad(D["dg2"], D["hid"], se["fc2"])
                ops.gemm_nt(D["dg2"], self._w(se["fc2"]["wt"], se["ph"], L["pe"]), _lib.EPI_BF16, out=D["dhid"])
                call("lafs_cnn_act_bwd_post", None, _p(D["dhid"]), _p(D["hid"]), D["dhid"].numel(), se["act1"], _p(D["dhid"]))
                self._wgrad(D["dhid"], D["pool"], se["fc1"])
                ops.gemm_nt(D["dhid"], self._w(se["fc1"]["wt"], L["pe"], se["ph"]), _lib.EPI_BF16, out=D["dpool"])
                dd_raw = view(b2, Ro, L["pe"])
                self._bn_bwd(L["bn2"], dz, D["d_raw"], Ro, _lib.ACT_NONE, dd_raw, add_nc=D["dpool"], HW=Ho * Ho)
                free_after_dw, keep = b2, b1
            else:
                self._bn_bwd(L["bn2"], dd, D["d_raw"], Ro, L["act"], dd_raw)
                free_after_dw, keep = b1, b2
            # depthwise: dd_raw [Ro, pe] -> de [R, pe]  (into `keep`, whose content is dead), dw accumulated in the arena
            if tr is not None:
                tr["dd_raw"] = dd_raw.clone()
            de = view(keep, R, L["pe"])
            call("lafs_cnn_dwconv_train_bwd", _p(D["e"]), _p(dd_raw), self._dww(L), N, H, H, L["pe"], L["cexp"], L["k"], L["stride"], _p(de),
                 self._dwg(L))
            if tr is not None:
                tr["de"] = de.clone()
            de_raw = view(free_after_dw, R, L["pe"])
            self._bn_bwd(L["bn1"], de, D["e_raw"], R, L["act"], de_raw)
            self._wgrad(de_raw, D["x_in"], L["exp"])
            dcur = view(keep, R, L["pi"])
            ops.gemm_nt(de_raw, self._w(L["exp"]["wt"], L["pi"], L["pe"]), _lib.EPI_BF16_ACT, out=dcur, aux=dy if L["residual"] else None,
                        act=_lib.ACT_NONE)
            if tr is not None:
                tr["dcur"] = dcur.clone()
                self.trace.append(tr)
            give(dy_buf); give(free_after_dw)
            dy_buf, dy = keep, dcur
        # stem
        H0 = self.S // 2
        R0 = N * H0 * H0
        st = self.stem
        b1 = take(); ds_raw = view(b1, R0, 32)
        self._bn_bwd(st["bn"], dy, B["s_raw"], R0, st["act"], ds_raw)
        self._wgrad(ds_raw, B["P"], st["conv"])
        # fold every padded weight gradient into the arena
        call("lafs_cnn_unpad_add_table", _p(self.gpad), _p(a.grad), _p(self.fold_table), _p(self.fold_starts), self.n_fold, self.fold_nblk,
             _p(self.gscale))
        # overflow guard: an inf / NaN that the 16-bit backward produced (BatchNorm rstd ~31 at eps 1e-3, squeeze-excite / depthwise
        # fan-in) must not reach AdamW -- the CNN's gradient range is checked and, if poisoned, zeroed; the scale target backs off
        call("lafs_cnn_grad_guard", self._g(self.grad_lo), self.grad_hi - self.grad_lo, _p(self.gscale), GRAD_TARGET)

    def overflow_state(self):
        """(current scale target, backwards dropped because of inf / NaN so far) -- one host sync; diagnostics and tests."""
        st = self.gscale.cpu().tolist()
        return st[2], int(st[4])

    def mark_stale(self):
        """The optimizer changed the master weights in place (no torch version bump): refresh the operand images next forward."""
        self._stale = True
