"""ConfigMixin / register_to_config stand-ins: record the constructor's keyword arguments (defaults included) as `self.config`."""
import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = "config.json"

    @property
    def config(self):
        return self._internal_dict

    def register_to_config(self, **kw):
        self._internal_dict = FrozenDict({**getattr(self, "_internal_dict", {}), **kw})

    @classmethod
    def from_config(cls, config, **kw):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")}, **kw)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)

    return inner
