"""Recursive weight containers: the tree an `*Infer` class walks as `weights.<name>.apply(...)`.

Public surface = the reference's (lightx2v/common/modules/weight_module.py:1-182), because its model code is written against it:
`add_module` / `register_parameter` attach a child under a name (also reachable as an attribute), `load(weight_dict)` configures
and loads every child, `state_dict` / `named_parameters` walk the tree (parameters before sub-modules, as the reference orders its
output), `to_cpu` / `to_cuda` (+ `_async`) and `clear` / `calculate_size` forward to the children that implement them.  Children
are duck-typed operator objects (`ops.py`), nothing here touches tensors.
"""

_PARAM, _MODULE = 0, 1


class WeightModule:
    def __init__(self):
        self._slots = []  # (kind, name, child) in attach order

    # ---- building the tree
    def _attach(self, kind, name, child):
        self._slots = [s for s in self._slots if s[1] != name]  # re-attaching a name replaces it
        self._slots.append((kind, name, child))
        setattr(self, name, child)
        return child

    def add_module(self, name, module):
        self._attach(_MODULE, name, module)

    def register_parameter(self, name, param):
        self._attach(_PARAM, name, param)

    def _walk(self, kinds=(_MODULE, _PARAM)):
        """Children that exist, sub-modules first unless asked otherwise (load order of the reference)."""
        for kind in kinds:
            for k, name, child in self._slots:
                if k == kind and child is not None:
                    yield name, child

    def _forward(self, method, *args, **kwargs):
        """Call `method` on every child that has it; returns the results."""
        return [getattr(child, method)(*args, **kwargs) for _, child in self._walk() if hasattr(child, method)]

    # ---- checkpoint in / out
    def load(self, weight_dict):
        for _, child in self._walk():
            if hasattr(child, "set_config"):
                child.set_config(self.config["mm_config"])
            if hasattr(child, "load"):
                child.load(weight_dict)

    def load_from_disk(self):
        """reference: weight_module.py:37-45 — the lazy-load path: every child that can re-read its tensors from its open checkpoint
        file does so (sub-modules first, then parameters)."""
        self._forward("load_from_disk")

    def state_dict(self, destination=None):
        destination = {} if destination is None else destination
        for _, child in self._walk((_PARAM, _MODULE)):
            child.state_dict(destination)
        return destination

    def named_parameters(self, prefix=""):
        for name, child in self._walk((_PARAM,)):
            yield prefix + name, child
        for name, child in self._walk((_MODULE,)):
            if hasattr(child, "named_parameters"):
                yield from child.named_parameters(f"{prefix}{name}.")
            else:  # a leaf operator object attached with add_module
                yield prefix + name, child

    # ---- lifecycle
    def calculate_size(self):
        return sum(self._forward("_calculate_size"))

    def clear(self):
        self._forward("clear")

    def to_cpu(self, non_blocking=False):
        self._forward("to_cpu", non_blocking=non_blocking)

    def to_cuda(self, non_blocking=False):
        self._forward("to_cuda", non_blocking=non_blocking)

    def to_cpu_async(self):
        self.to_cpu(non_blocking=True)

    def to_cuda_async(self):
        self.to_cuda(non_blocking=True)


class WeightModuleList(WeightModule):
    """An indexable sequence of sub-modules; child i is attached under the name str(i) (checkpoint names `blocks.{i}. ...`)."""

    def __init__(self, modules=None):
        super().__init__()
        self._seq = []
        for module in modules or ():
            self.append(module)

    def append(self, module):
        self.add_module(str(len(self._seq)), module)
        self._seq.append(module)

    def load(self, weight_dict):  # list members carry their own config (no set_config pass at this level)
        for module in self._seq:
            module.load(weight_dict)

    def __getitem__(self, idx):
        return self._seq[idx]

    def __len__(self):
        return len(self._seq)

    def __iter__(self):
        return iter(self._seq)
