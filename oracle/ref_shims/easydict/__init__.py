"""Minimal stand-in for `easydict.EasyDict` (attribute access on a dict), enough for the reference's
config objects. Test infrastructure only."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v
