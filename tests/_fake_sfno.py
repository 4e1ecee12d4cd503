"""TEST INFRASTRUCTURE: an emulation of the SFNO side of the C ABI (ace_sfno_create / set_weight / forward[_graph] /
forward_conditioned / weights_generation / destroy, ace_pack_normalize, ace_unpack_denormalize; include/ace_sfno.h) on host memory - the forward is the
CPU oracle network (oracle/sfno.py) built from the uploaded weights, pack / unpack walk the same device pointer tables and strides -
so that the HOST logic of the rollout engine (static buffers, per-step pointer tables, forcing indices, hooks, derived forcings,
window feeder) runs in the ``-m "not gpu"`` suite.  Not a fallback: nothing in the product imports it and the product's own
entry points still refuse host tensors; the GPU tests run the real kernels.  Use: ``with fake_sfno(): RolloutEngine(...)``."""
import contextlib
import ctypes
import os

import numpy as np
import torch


def _view(ptr, n, ctype):
    return torch.from_numpy(np.ctypeslib.as_array((ctype * int(n)).from_address(int(ptr))))


def _f32(ptr, n):
    return _view(ptr, n, ctypes.c_float)


class _Net:
    def __init__(self, cfg):
        from oracle.sfno import SFNOConfig, init_state
        import ace_amd.sfno as S
        inv = lambda d: {v: k for k, v in d.items()}          # noqa: E731
        self.cfg = SFNOConfig(
            in_chans=cfg.in_chans, out_chans=cfg.out_chans, img_shape=(cfg.nlat, cfg.nlon), operator_type=inv(S._OPERATOR)[cfg.operator_type],
            scale_factor=cfg.scale_factor, embed_dim=cfg.embed_dim, num_layers=cfg.num_layers,
            hard_thresholding_fraction=cfg.hard_thresholding_fraction, normalization_layer=inv(S._NORM)[cfg.normalization_layer],
            use_mlp=bool(cfg.use_mlp), mlp_ratio=cfg.mlp_ratio, activation_function=inv(S._ACT)[cfg.activation_function],
            encoder_layers=cfg.encoder_layers, pos_embed=bool(cfg.pos_embed), big_skip=bool(cfg.big_skip), data_grid=inv(S._GRID)[cfg.data_grid])
        self.shapes = {k: tuple(v.shape) for k, v in init_state(self.cfg, seed=0).items()}
        self.state = {}
        self.generation = 0
        self.oracle = None
        self.forwards = 0

    def forward(self, x):
        from oracle.sfno import SFNOOracle
        if self.oracle is None:
            missing = set(self.shapes) - set(self.state)
            assert not missing, f"weights never uploaded: {sorted(missing)}"
            self.oracle = SFNOOracle(self.cfg, self.state, dtype=torch.float32)
        self.forwards += 1
        return self.oracle.forward(x)


class _CNet:
    """the conditional network (normalization_layer == 2): the CPU oracle of NoiseConditionedSFNO with ONE conditioning field of
    `noise_embed_dim` channels - exactly what the native network sees (the host merges noise / positional / label context)"""

    def __init__(self, cfg):
        from oracle.csfno import CSFNOConfig
        import ace_amd
        import ace_amd.sfno as S
        inv = lambda d: {v: k for k, v in d.items()}          # noqa: E731
        kw = dict(embed_dim=cfg.embed_dim, noise_embed_dim=cfg.noise_embed_dim, num_layers=cfg.num_layers, use_mlp=bool(cfg.use_mlp),
                  mlp_ratio=cfg.mlp_ratio, activation_function=inv(S._ACT)[cfg.activation_function], encoder_layers=cfg.encoder_layers,
                  pos_embed=bool(cfg.pos_embed), big_skip=bool(cfg.big_skip), data_grid=inv(S._GRID)[cfg.data_grid],
                  normalize_big_skip=bool(cfg.normalize_big_skip), affine_norms=bool(cfg.affine_norms), filter_num_groups=cfg.filter_num_groups)
        self.cfg = CSFNOConfig(in_chans=cfg.in_chans, out_chans=cfg.out_chans, img_shape=(cfg.nlat, cfg.nlon), noise_type="gaussian", **kw)
        template = ace_amd.ModuleSelector(type="NoiseConditionedSFNO", config=kw).build(
            cfg.in_chans, cfg.out_chans, ace_amd.DatasetInfo((cfg.nlat, cfg.nlon))).torch_module.conditional_model
        self.shapes = {k: tuple(v.shape) for k, v in template.state_dict().items()}
        self.state = {}
        self.generation = 0
        self.oracle = None
        self.forwards = 0

    def forward(self, x, cond):
        from oracle.csfno import CSFNOOracle
        if self.oracle is None:
            missing = set(self.shapes) - set(self.state)
            assert not missing, f"weights never uploaded: {sorted(missing)}"
            self.oracle = CSFNOOracle(self.cfg, self.state, dtype=torch.float32)
        self.forwards += 1
        return self.oracle.forward(x, noise=cond)


class FakeSfno:
    ACE_OK, ACE_ERR_INVALID = 0, 1

    def __init__(self, real):
        self._real = real
        self._nets = {}
        self._next = 1
        self.AceSfnoConfig = real.AceSfnoConfig

    def __getattr__(self, name):           # structs / constants of the real binding module
        return getattr(self._real, name)

    # -- what the product reads from the _lib module
    def lib(self):
        return self

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"emulated C ABI returned {rc}")

    @staticmethod
    def ptr(t):
        return None if t is None else t.data_ptr()

    @staticmethod
    def current_stream():
        return None

    # -- entry points
    def ace_sfno_create(self, cfg, out):
        self._nets[self._next] = _CNet(cfg._obj) if cfg._obj.normalization_layer == 2 else _Net(cfg._obj)
        out._obj.value = self._next
        self._next += 1
        return 0

    def ace_sfno_destroy(self, h):
        self._nets.pop(getattr(h, "value", h), None)

    def _net(self, h):
        return self._nets[getattr(h, "value", h)]

    def ace_sfno_set_weight(self, h, name, ptr, numel, stream):
        net = self._net(h)
        name = name.decode()
        shape = net.shapes[name]
        assert int(np.prod(shape)) == numel, (name, shape, numel)
        net.state[name] = _f32(ptr, numel).clone().reshape(shape)
        net.oracle = None
        net.generation += 1
        return 0

    def ace_sfno_weights_generation(self, h):
        return self._net(h).generation

    def ace_sfno_forward(self, h, x, y, batch, stream):
        net = self._net(h)
        c = net.cfg
        hw = c.img_shape[0] * c.img_shape[1]
        xin = _f32(x, batch * c.in_chans * hw).view(batch, c.in_chans, *c.img_shape)
        _f32(y, batch * c.out_chans * hw).view(batch, c.out_chans, *c.img_shape).copy_(net.forward(xin))
        return 0

    ace_sfno_forward_graph = ace_sfno_forward

    def ace_sfno_forward_conditioned(self, h, x, cond, y, batch, stream):
        net = self._net(h)
        c = net.cfg
        hw = c.img_shape[0] * c.img_shape[1]
        xin = _f32(x, batch * c.in_chans * hw).view(batch, c.in_chans, *c.img_shape)
        field = _f32(cond, batch * c.noise_embed_dim * hw).view(batch, c.noise_embed_dim, *c.img_shape)
        _f32(y, batch * c.out_chans * hw).view(batch, c.out_chans, *c.img_shape).copy_(net.forward(xin, field))
        return 0

    def ace_pack_normalize(self, srcs, strides, mean, std, dst, batch, nch, hw, stream):
        p = _view(srcs, nch, ctypes.c_int64)
        s = _view(strides, nch, ctypes.c_int64)
        m, d = _f32(mean, nch), _f32(std, nch)
        out = _f32(dst, batch * nch * hw).view(batch, nch, hw)
        for c in range(nch):
            src = torch.as_strided(_f32(int(p[c]), (batch - 1) * int(s[c]) + hw), (batch, hw), (int(s[c]), 1))
            out[:, c] = (src - m[c]) / d[c]
        return 0

    def ace_unpack_denormalize(self, src, mean, std, dsts, strides, batch, nch, hw, stream):
        p = _view(dsts, nch, ctypes.c_int64)
        s = _view(strides, nch, ctypes.c_int64)
        m, d = _f32(mean, nch), _f32(std, nch)
        x = _f32(src, batch * nch * hw).view(batch, nch, hw)
        for c in range(nch):
            dst = torch.as_strided(_f32(int(p[c]), (batch - 1) * int(s[c]) + hw), (batch, hw), (int(s[c]), 1))
            dst.copy_(x[:, c] * d[c] + m[c])
        return 0


@contextlib.contextmanager
def fake_sfno():
    """ace_amd.{sfno, rollout} bound to the emulation; the post-step hooks run as torch ops (ACE_NO_FUSED_PHYSICS)."""
    import ace_amd.csfno as C
    import ace_amd.rollout as R
    import ace_amd.sfno as S
    from ace_amd import _lib as real
    fake = FakeSfno(real)
    saved = (S._lib, R._lib, torch.cuda.device, os.environ.get("ACE_NO_FUSED_PHYSICS"), C._lib)
    S._lib = R._lib = C._lib = fake
    torch.cuda.device = lambda device: contextlib.nullcontext()
    os.environ["ACE_NO_FUSED_PHYSICS"] = "1"
    try:
        yield fake
    finally:
        S._lib, R._lib, torch.cuda.device, C._lib = saved[0], saved[1], saved[2], saved[4]
        if saved[3] is None:
            os.environ.pop("ACE_NO_FUSED_PHYSICS", None)
        else:
            os.environ["ACE_NO_FUSED_PHYSICS"] = saved[3]
