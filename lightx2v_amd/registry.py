"""String-keyed operator registries — the drop-in boundary of the reference
(reference: lightx2v/utils/registry_factory.py:1-56).  Same registry names and the same decorator
protocol (`@MM_WEIGHT_REGISTER("key")`, duplicate key raises), so config strings such as
`mm_config.mm_type`, `self_attn_1_type`, `cross_attn_1_type` select our operators the way they select
the reference's.  `lightx2v_amd.plugin.register_into_reference()` adds the same classes to the
reference's own registry objects when LightX2V itself is importable.
"""


class Register(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._dict = {}

    def __call__(self, target_or_name):
        if callable(target_or_name):
            return self.register(target_or_name)
        return lambda x: self.register(x, key=target_or_name)

    def register(self, target, key=None):
        if not callable(target):
            raise Exception(f"Error: {target} must be callable!")
        if key is None:
            key = target.__name__
        if key in self._dict:
            raise Exception(f"{key} already exists.")
        self[key] = target
        return target

    def __setitem__(self, key, value):
        self._dict[key] = value

    def __getitem__(self, key):
        return self._dict[key]

    def __contains__(self, key):
        return key in self._dict

    def __str__(self):
        return str(self._dict)

    def keys(self):
        return self._dict.keys()

    def values(self):
        return self._dict.values()

    def items(self):
        return self._dict.items()


MM_WEIGHT_REGISTER = Register()
ATTN_WEIGHT_REGISTER = Register()
RMS_WEIGHT_REGISTER = Register()
LN_WEIGHT_REGISTER = Register()
CONV3D_WEIGHT_REGISTER = Register()
TENSOR_REGISTER = Register()
