"""TEST INFRASTRUCTURE ONLY.  A small stand-in for the third-party ``dacite`` package (absent from this image) so that
the reference's own config classes can be built from dicts inside the build container (oracle/ref_loader.py).  Written
from dacite's documented behaviour (``from_dict(data_class, data, config=Config(strict=...))``: nested dataclasses,
Optional / Union members tried in order, typed lists / dicts / tuples, Literal checks, strict = unknown keys raise);
nothing in ace_amd/ imports it."""
import collections.abc as cabc
import dataclasses
import types
import typing


class DaciteError(Exception):
    pass


class UnexpectedDataError(DaciteError):
    pass


class WrongTypeError(DaciteError):
    pass


class MissingValueError(DaciteError):
    pass


class UnionMatchError(WrongTypeError):
    pass


class Config:
    def __init__(self, strict=False, **kw):
        self.strict = strict


def _coerce(tp, v, config):
    if tp is typing.Any:
        return v
    origin = typing.get_origin(tp)
    if origin in (typing.Union, types.UnionType):
        args = typing.get_args(tp)
        if v is None and type(None) in args:
            return None
        errs = []
        for a in args:
            if a is type(None):
                continue
            try:
                return _coerce(a, v, config)
            except DaciteError as e:
                errs.append(e)
        raise UnionMatchError(f"no member of {tp} matches {type(v).__name__}: {errs}")
    if origin is typing.Literal:
        if v not in typing.get_args(tp):
            raise WrongTypeError(f"{v!r} not in {tp}")
        return v
    if dataclasses.is_dataclass(tp) and isinstance(tp, type):
        if isinstance(v, tp):
            return v
        if isinstance(v, cabc.Mapping):
            return from_dict(tp, v, config)
        raise WrongTypeError(f"expected {tp.__name__}, got {type(v).__name__}")
    if origin in (list, cabc.Sequence, cabc.MutableSequence, tuple, set):
        if isinstance(v, (str, bytes)) or not isinstance(v, (list, tuple, set)):
            raise WrongTypeError(f"expected {tp}, got {type(v).__name__}")
        args = typing.get_args(tp)
        if origin is tuple and args and args[-1] is not Ellipsis:
            return tuple(_coerce(a, x, config) for a, x in zip(args, v))
        inner = args[0] if args else typing.Any
        seq = [_coerce(inner, x, config) for x in v]
        return tuple(seq) if origin is tuple else (set(seq) if origin is set else seq)
    if origin in (dict, cabc.Mapping, cabc.MutableMapping):
        if not isinstance(v, cabc.Mapping):
            raise WrongTypeError(f"expected {tp}, got {type(v).__name__}")
        args = typing.get_args(tp)
        inner = args[1] if len(args) == 2 else typing.Any
        return {k: _coerce(inner, x, config) for k, x in v.items()}
    if isinstance(tp, type):
        if tp is float and isinstance(v, int) and not isinstance(v, bool):
            return v
        if not isinstance(v, tp):
            raise WrongTypeError(f"expected {tp.__name__}, got {type(v).__name__} ({v!r})")
    return v


def from_dict(data_class, data, config=None):
    config = config or Config()
    hints = typing.get_type_hints(data_class)
    fields = {f.name: f for f in dataclasses.fields(data_class) if f.init}
    if config.strict:
        extra = set(data) - set(fields)
        if extra:
            raise UnexpectedDataError(f"{data_class.__name__}: unexpected keys {sorted(extra)}")
    kwargs = {}
    for name, f in fields.items():
        if name in data:
            kwargs[name] = _coerce(hints.get(name, typing.Any), data[name], config)
        elif f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING:
            raise MissingValueError(f"{data_class.__name__}: missing {name}")
    return data_class(**kwargs)
