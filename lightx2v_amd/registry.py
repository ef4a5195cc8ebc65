"""String-keyed operator registries: how a JSON config string (`mm_config.mm_type`, `self_attn_1_type`, `cross_attn_1_type`, ...)
picks an operator class.  The protocol is the one the reference's model code is written against
(lightx2v/utils/registry_factory.py:1-56): `@REG("key")` or bare `@REG` registers a class, registering a taken key through the
decorator is an error, `REG["key"]` looks a class up, and plain item assignment installs an alias (the reference's weight
classes hard-code a few keys — `LN_WEIGHT_REGISTER["Default"]`, `RMS_WEIGHT_REGISTER["sgl-kernel"]` — which `ops.py` aliases
onto the HIP classes this way).  `lightx2v_amd.plugin.register_into_reference()` puts the same classes into the reference's
own registry objects when LightX2V itself is importable.
"""
from collections.abc import Mapping


class DuplicateKey(Exception):
    """Raised by the decorator path for a key that is already bound (the reference raises a bare Exception here)."""


class Registry(Mapping):
    """name -> factory table with a decorator front end.  Read access is the Mapping interface; writes go through the decorator
    (`reg("key")(cls)` / `reg(cls)`, refusing duplicates) or through `reg["key"] = cls` (an explicit alias / override)."""

    __slots__ = ("kind", "_table")

    def __init__(self, kind):
        self.kind = kind
        self._table = {}

    # ---- writes
    def _bind(self, key, factory, overwrite):
        if not callable(factory):
            raise TypeError(f"{self.kind} registry: {factory!r} is not callable")
        if not overwrite and key in self._table:
            raise DuplicateKey(f"{self.kind} registry: key {key!r} is already bound to {self._table[key].__name__}")
        self._table[key] = factory
        return factory

    def register(self, factory, key=None):
        return self._bind(factory.__name__ if key is None else key, factory, overwrite=False)

    def __call__(self, key_or_factory):
        if isinstance(key_or_factory, str):
            return lambda factory: self._bind(key_or_factory, factory, overwrite=False)
        return self.register(key_or_factory)

    def __setitem__(self, key, factory):
        self._bind(key, factory, overwrite=True)

    # ---- reads (Mapping supplies keys / values / items / get / __contains__)
    def __getitem__(self, key):
        try:
            return self._table[key]
        except KeyError:
            raise KeyError(f"{self.kind} registry has no {key!r}; known: {sorted(self._table)}") from None

    def __iter__(self):
        return iter(self._table)

    def __len__(self):
        return len(self._table)

    def __repr__(self):
        return f"Registry({self.kind}: {', '.join(sorted(self._table))})"


MM_WEIGHT_REGISTER = Registry("mm_weight")
ATTN_WEIGHT_REGISTER = Registry("attn_weight")
RMS_WEIGHT_REGISTER = Registry("rms_weight")
LN_WEIGHT_REGISTER = Registry("ln_weight")
CONV3D_WEIGHT_REGISTER = Registry("conv3d_weight")
TENSOR_REGISTER = Registry("tensor")
