"""TEST INFRASTRUCTURE: an emulation of the ``ace_hpx_*`` C-ABI operators (include/ace_sfno.h, csrc/healpix.hip) on host memory,
in plain torch fp32 / fp64 arithmetic, so that the HOST logic of ace_amd/healpix.py - layouts, pitches, padding tables, row-offset
tables, weight preparation order, block composition, skip handling - and of ace_amd/mlp.py (the "MLP" registry network on the
k = 1 operator) can be checked against the reference's golden outputs in
the ``-m "not gpu"`` suite.  It restates what each entry point is documented to compute (the comments above each function in
healpix.hip); it is NOT a fallback: nothing in the product imports it, ``HEALPixUNet.forward`` still refuses host tensors, and the
GPU tests run the real kernels.  Use: ``with fake_hpx(): net._run(x)``."""
import contextlib
import ctypes
import types

import numpy as np
import torch


def _view(ptr, n, ctype, dtype):
    if n <= 0:
        return torch.zeros(0, dtype=dtype)
    return torch.from_numpy(np.ctypeslib.as_array((ctype * int(n)).from_address(int(ptr))))


def _f32(ptr, n):
    return _view(ptr, n, ctypes.c_float, torch.float32)


def _act(v, act, cap):
    if act == 1:
        v = 0.5 * v * (1.0 + torch.erf(v * 0.70710678118654752440))
    elif act == 2:
        v = torch.clamp(v, min=0.0)
    return torch.clamp(v, max=cap)


class FakeHpx:
    ACE_ERR_INVALID = 1

    def __init__(self, real_lib):
        self._real = real_lib
        self._weights = {}
        self._next = 1
        self.calls = []

    # -- what ace_amd.healpix reads from the _lib module
    def lib(self):
        return self

    @staticmethod
    def ptr(t):
        return None if t is None else t.data_ptr()

    @staticmethod
    def current_stream():
        return None

    # -- entry points
    def ace_hpx_last_error(self):
        return b"(emulated ace_hpx)"

    def ace_hpx_pad_table_host(self, nside, p, ia, ib):
        return self._real.ace_hpx_pad_table_host(nside, p, ia, ib)          # host function of the real library

    def ace_hpx_absmax(self, x, n, amax, stream):
        self.calls.append("absmax")
        return 0

    def ace_hpx_weight_create(self, w_dev, rows, cols, stream, out):
        self._weights[self._next] = _f32(w_dev, rows * cols).clone().reshape(rows, cols).double()
        out._obj.value = self._next
        self._next += 1
        return 0

    def ace_hpx_weight_destroy(self, h):
        self._weights.pop(getattr(h, "value", h), None)

    def _w(self, h):
        return self._weights[getattr(h, "value", h)]

    def ace_hpx_pad(self, x, x_img_stride, x_chan_stride, x_pitch, y, y_chans, c0, c, ia, ib, items, nside, p, y_pitch, amax, stream):
        self.calls.append("pad")
        m = nside + 2 * p
        if y_pitch < m or p < 1 or c0 + c > y_chans:
            return 1
        a = _view(ia, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        b = _view(ib, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        imgs = items * 12
        top = (imgs - 1) * x_img_stride + (c - 1) * x_chan_stride + 4095 * 0
        src = _f32(x, top + int(max(((a >> 12) & 4095).max(), ((b >> 12) & 4095).max())) * x_pitch + int(max((a & 4095).max(), (b & 4095).max())) + 1)
        cells = m * y_pitch
        dst = _f32(y, imgs * y_chans * cells + 16)
        out = dst[: imgs * y_chans * cells].view(imgs, y_chans, m, y_pitch)
        item = torch.arange(items).view(items, 1, 1, 1, 1)
        ch = torch.arange(c).view(1, 1, c, 1, 1)

        def gather(s):
            s = s.view(1, 12, 1, m, m)
            return src[((item * 12 + (s >> 24)) * x_img_stride + ch * x_chan_stride + ((s >> 12) & 4095) * x_pitch + (s & 4095))]

        va, vb = gather(a), gather(b)
        same = (a == b).view(1, 12, 1, m, m)
        v = torch.where(same, va, 0.5 * va + 0.5 * vb).reshape(imgs, c, m, m)
        out[:, c0:c0 + c, :, :m] = v
        out[:, c0:c0 + c, :, m:] = 0.0
        if c0 + c == y_chans:
            dst[imgs * y_chans * cells:] = 0.0
        return 0

    def ace_hpx_conv(self, x, x2, cin, cin2, w, row_off, bias, R, y, imgs, cout, H, W, pitch, k, dil, act, cap, xmax, x2max, ymax, stream):
        self.calls.append(f"conv{k}")
        Wm = self._w(w)
        K = (cin + cin2) * k * k
        if Wm.shape != (cout, K) or pitch % 4 or pitch < W + (k - 1) * dil or (k > 1 and (x2 or R)):
            return 1
        rows_in = H + (k - 1) * dil
        N = H * pitch
        sB = cin * rows_in * pitch
        xs = _f32(x, imgs * sB + (16 if k > 1 else 0)).double()
        if k > 1:
            off = _view(row_off, K, ctypes.c_int64, torch.int64)
            col = torch.arange(N)
            B = torch.stack([xs[i * sB + off[:, None] + col[None, :]] for i in range(imgs)])       # [imgs][K][N]
        else:
            B = xs[: imgs * sB].view(imgs, cin, N)
            if cin2:
                B = torch.cat([B, _f32(x2, imgs * cin2 * N).double().view(imgs, cin2, N)], dim=1)
        out = torch.einsum("ok,ikn->ion", Wm, B)
        if bias:
            out = out + _f32(bias, cout).double().view(1, cout, 1)
        if R:
            out = out + _f32(R, imgs * cout * N).double().view(imgs, cout, N)
        _f32(y, imgs * cout * N).view(imgs, cout, N).copy_(_act(out, act, cap).float())
        return 0

    def ace_hpx_pad_planes(self, x, x_img_stride, x_chan_stride, x_pitch, x2, x2_img_stride, x2_chan_stride, x2_pitch, cin, cin2, hi, lo, ia, ib,
                           items, nside, p, y_pitch, xmax, x2max, pmax, stream):
        """the gather of ace_hpx_pad for both sources, stored as documented: fp16 hi / lo planes [img][channel group][cell][8] (the
        emulated bound slots are zero: scale 2^0)"""
        self.calls.append("pad_planes")
        m = nside + 2 * p
        if y_pitch < m or p < 1 or (cin2 and not x2):
            return 1
        a = _view(ia, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        b = _view(ib, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        imgs, ctot = items * 12, cin + cin2
        cg8, cells = (ctot + 7) // 8, m * y_pitch
        ymax, xmax_ = int(max(((a >> 12) & 4095).max(), ((b >> 12) & 4095).max())), int(max((a & 4095).max(), (b & 4095).max()))
        item = torch.arange(items).view(items, 1, 1, 1, 1)
        same = (a == b).view(1, 12, 1, m, m)
        full = torch.zeros(imgs, cg8 * 8, m, y_pitch, dtype=torch.float32)
        for ptr, ist, cst, pt, c, c0 in ((x, x_img_stride, x_chan_stride, x_pitch, cin, 0), (x2, x2_img_stride, x2_chan_stride, x2_pitch, cin2, cin)):
            if not c:
                continue
            src = _f32(ptr, (imgs - 1) * ist + (c - 1) * cst + ymax * pt + xmax_ + 1)
            ch = torch.arange(c).view(1, 1, c, 1, 1)

            def gather(s_):
                s_ = s_.view(1, 12, 1, m, m)
                return src[((item * 12 + (s_ >> 24)) * ist + ch * cst + ((s_ >> 12) & 4095) * pt + (s_ & 4095))]

            va, vb = gather(a), gather(b)
            full[:, c0:c0 + c, :, :m] = torch.where(same, va, 0.5 * va + 0.5 * vb).reshape(imgs, c, m, m)
        ent = full.view(imgs, cg8, 8, cells).permute(0, 1, 3, 2).contiguous()        # [img][cg][cell][8]
        h16 = ent.to(torch.float16)
        l16 = (ent - h16.float()).to(torch.float16)
        n = imgs * cg8 * cells * 8
        for ptr, val in ((hi, h16), (lo, l16)):
            dst = _view(ptr, n + 16 * 8, ctypes.c_uint16, torch.int16)
            dst[:n] = val.reshape(-1).view(torch.int16)
            dst[n:] = 0
        return 0

    def ace_hpx_conv_packed(self, xhi, xlo, cpad, x_plane_cells, w, bias, bias_max, y, yhi, ylo, y_plane_cells, imgs, cout, H, W, pitch, k, dil,
                            act, cap, pmax, ymax, stream):
        """xhi / yhi may point at a shifted origin inside larger padded planes (x_plane_cells / y_plane_cells = their entries per plane)"""
        self.calls.append(f"conv{k}" + ("<-interior" if k == 1 else "") + ("->planes" if yhi and not y_plane_cells else "") + ("->padded" if y_plane_cells else ""))
        Wm = self._w(w)
        K = cpad * k * k
        if Wm.shape != (cout, K) or pitch % 4 or pitch < W + (k - 1) * dil or cpad % 8 or k < 1:
            return 1
        rows_in, cg8 = H + (k - 1) * dil, cpad // 8
        cells, N = (x_plane_cells or rows_in * pitch), H * pitch
        n = (imgs * cg8 - 1) * cells * 8 + (((k - 1) * pitch + (k - 1)) * dil + N) * 8
        xs = (_view(xhi, n, ctypes.c_uint16, torch.int16).view(torch.float16).double() + _view(xlo, n, ctypes.c_uint16, torch.int16).view(torch.float16).double())
        ent = xs.view(-1, 8)                                                              # [entry][8 channels]
        col = torch.arange(N)
        out = torch.zeros(imgs, cout, N, dtype=torch.float64)
        Wt = Wm.view(cout, k * k, cg8, 8)
        for t in range(k * k):
            toff = ((t // k) * pitch + (t % k)) * dil
            for cg in range(cg8):
                for i in range(imgs):
                    Bt = ent[(i * cg8 + cg) * cells + toff + col]                         # [N][8]
                    out[i] += Wt[:, t, cg, :] @ Bt.T
        if bias:
            out = out + _f32(bias, cout).double().view(1, cout, 1)
        res = _act(out, act, cap).float()
        if y:
            _f32(y, imgs * cout * N).view(imgs, cout, N).copy_(res)
        if yhi:                                                                           # [img][cout / 8][plane cells][8], emulated scale 2^0
            if cout % 8:
                return 1
            ycells = y_plane_cells or N
            ent_o = res.view(imgs, cout // 8, 8, N).permute(0, 1, 3, 2).contiguous()      # [img][cg][N][8]
            h16 = ent_o.to(torch.float16)
            l16 = (ent_o - h16.float()).to(torch.float16)
            span = (imgs * (cout // 8) - 1) * ycells * 8 + N * 8
            for ptr, val in ((yhi, h16), (ylo, l16)):
                dst = torch.as_strided(_view(ptr, span, ctypes.c_uint16, torch.int16), (imgs, cout // 8, N, 8), ((cout // 8) * ycells * 8, ycells * 8, 8, 1))
                dst.copy_(val.view(torch.int16))
        return 0

    def ace_hpx_halo_planes(self, hi, lo, cpad, ia, ib, items, nside, p, y_pitch, stream):
        """halo cells gathered in place from the interiors of the neighbouring faces' planes; gap columns and slack zeroed"""
        self.calls.append("halo")
        m = nside + 2 * p
        if y_pitch < m or p < 1 or cpad % 8:
            return 1
        a = _view(ia, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        b = _view(ib, 12 * m * m, ctypes.c_int32, torch.int32).long().reshape(12, m, m)
        imgs, cg8, cells = items * 12, cpad // 8, m * y_pitch
        n = imgs * cg8 * cells * 8
        bufs = [_view(ptr, n + 16 * 8, ctypes.c_uint16, torch.int16) for ptr in (hi, lo)]
        val = sum(buf[:n].view(torch.float16).double() for buf in bufs).view(items, 12, cg8, m, y_pitch, 8)
        interior = val[:, :, :, p:p + nside, p:p + nside, :]                              # [item][face][cg][y][x][8]

        def gather(s_):                                                                   # -> [item][12][cg][m][m][8]
            src = interior.permute(1, 3, 4, 0, 2, 5)                                      # [face][y][x][item][cg][8]
            return src[s_ >> 24, (s_ >> 12) & 4095, s_ & 4095].permute(3, 0, 4, 1, 2, 5)  # [12][m][m][item][cg][8] -> target order

        g = 0.5 * gather(a) + 0.5 * gather(b)
        out = torch.zeros_like(val)
        out[:, :, :, :, :m, :] = g
        out[:, :, :, p:p + nside, p:p + nside, :] = interior
        flat = out.reshape(-1).float()
        h16 = flat.to(torch.float16)
        l16 = (flat - h16.float()).to(torch.float16)
        keep = torch.zeros(items, 12, cg8, m, y_pitch, 8, dtype=torch.bool)
        keep[:, :, :, p:p + nside, p:p + nside, :] = True                                 # interiors keep their bits
        keep = keep.reshape(-1)
        for buf, v16 in zip(bufs, (h16, l16)):
            b16 = buf.view(torch.int16)
            b16[:n] = torch.where(keep, b16[:n], v16.view(torch.int16))
            b16[n:] = 0
        return 0

    def ace_hpx_conv1_packed(self, xhi, xlo, cin, w, bias, R, y, imgs, cout, H, W, pitch, act, xslot, ymax, stream):
        self.calls.append("conv1<-planes")
        Wm = self._w(w)
        if Wm.shape != (cout, cin) or pitch % 4 or cin % 8:
            return 1
        N = H * pitch
        n = imgs * cin * N
        xs = (_view(xhi, n, ctypes.c_uint16, torch.int16).view(torch.float16).double() + _view(xlo, n, ctypes.c_uint16, torch.int16).view(torch.float16).double())
        B = xs.view(imgs, cin // 8, N, 8).permute(0, 1, 3, 2).reshape(imgs, cin, N)
        out = torch.einsum("ok,ikn->ion", Wm, B)
        if bias:
            out = out + _f32(bias, cout).double().view(1, cout, 1)
        if R:
            out = out + _f32(R, imgs * cout * N).double().view(imgs, cout, N)
        _f32(y, imgs * cout * N).view(imgs, cout, N).copy_(_act(out, act, float("inf")).float())
        return 0

    def ace_hpx_pool2(self, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out, is_max, stream):
        self.calls.append("pool")
        if H % 2 or W % 2:
            return 1
        src = _f32(x, (planes - 1) * plane_stride_in + H * pitch_in)
        dst = _f32(y, (planes - 1) * plane_stride_out + (H // 2) * pitch_out)
        s = torch.as_strided(src, (planes, H, W), (plane_stride_in, pitch_in, 1))
        q = torch.stack([s[:, 0::2, 0::2], s[:, 0::2, 1::2], s[:, 1::2, 0::2], s[:, 1::2, 1::2]])
        r = q.max(dim=0).values if is_max else (((q[0] + q[1]) + q[2]) + q[3]) * 0.25
        torch.as_strided(dst, (planes, H // 2, W // 2), (plane_stride_out, pitch_out, 1)).copy_(r)
        return 0

    def ace_hpx_upsample2(self, x, y, planes, H, W, pitch_in, plane_stride_in, pitch_out, plane_stride_out, mode, align_corners, stream):
        self.calls.append("upsample")
        src = _f32(x, (planes - 1) * plane_stride_in + H * pitch_in)
        dst = _f32(y, (planes - 1) * plane_stride_out + 2 * H * pitch_out)
        s = torch.as_strided(src, (1, planes, H, W), (0, plane_stride_in, pitch_in, 1))
        if mode == 0:
            r = torch.nn.functional.interpolate(s, scale_factor=2, mode="nearest")
        else:
            r = torch.nn.functional.interpolate(s, scale_factor=2, mode="bilinear", align_corners=bool(align_corners))
        torch.as_strided(dst, (planes, 2 * H, 2 * W), (plane_stride_out, pitch_out, 1)).copy_(r[0])
        return 0

    def ace_hpx_tconv2(self, x, w, bias, tmp, y, imgs, cin, cout, H, W, pitch_in, pitch_out, plane_stride_out, act, cap, xmax, ymax, stream):
        self.calls.append("tconv")
        Wm = self._w(w)
        if Wm.shape != (4 * cout, cin) or pitch_in % 4:
            return 1
        N = H * pitch_in
        B = _f32(x, imgs * cin * N).double().view(imgs, cin, N)
        t = torch.einsum("ok,ikn->ion", Wm, B).view(imgs, 4, cout, H, pitch_in)
        if bias:
            t = t + _f32(bias, cout).double().view(1, 1, cout, 1, 1)
        t = _act(t, act, cap).float()
        dst = _f32(y, (imgs * cout - 1) * plane_stride_out + 2 * H * pitch_out)
        out = torch.as_strided(dst, (imgs, cout, 2 * H, 2 * W), (cout * plane_stride_out, plane_stride_out, pitch_out, 1))
        for dy in (0, 1):
            for dx in (0, 1):
                out[:, :, dy::2, dx::2] = t[:, dy * 2 + dx, :, :, :W]
        return 0


@contextlib.contextmanager
def fake_hpx():
    """ace_amd.healpix bound to the emulation for the duration of the block (its runtime caches are reset on both sides)."""
    import ace_amd.healpix as hp
    import ace_amd.mlp as mlp
    from ace_amd import _lib as real
    fake = FakeHpx(real.lib())
    saved = (hp._lib, mlp._lib, mlp.ColumnMLP.forward)
    hp._lib = mlp._lib = fake
    mlp.ColumnMLP.forward = mlp.ColumnMLP._run          # the body without the "must be on the GPU" guard
    hp._RT.__init__()
    hp._IDENTITY.clear()
    try:
        yield fake
    finally:
        hp._lib, mlp._lib, mlp.ColumnMLP.forward = saved
        hp._RT.__init__()
        hp._IDENTITY.clear()
