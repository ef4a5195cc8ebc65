"""Recursive weight containers (reference: lightx2v/common/modules/weight_module.py:1-182): `load`
hands every child the checkpoint dict, `to_cpu/to_cuda` move them, `state_dict` re-exports.  Children are
duck-typed operator objects (`load/apply/to_cuda/to_cpu/state_dict`)."""


class WeightModule:
    def __init__(self):
        self._modules = {}
        self._parameters = {}

    def add_module(self, name, module):
        self._modules[name] = module
        setattr(self, name, module)

    def register_parameter(self, name, param):
        self._parameters[name] = param
        setattr(self, name, param)

    def _children(self):
        yield from self._modules.values()
        yield from self._parameters.values()

    def load(self, weight_dict):
        for child in self._children():
            if hasattr(child, "set_config"):
                child.set_config(self.config["mm_config"])
            if hasattr(child, "load"):
                child.load(weight_dict)

    def calculate_size(self):
        return sum(c._calculate_size() for c in self._children() if hasattr(c, "_calculate_size"))

    def clear(self):
        for child in self._children():
            if hasattr(child, "clear"):
                child.clear()

    def state_dict(self, destination=None):
        if destination is None:
            destination = {}
        for child in list(self._parameters.values()) + list(self._modules.values()):
            if child is not None:
                child.state_dict(destination)
        return destination

    def named_parameters(self, prefix=""):
        for name, param in self._parameters.items():
            if param is not None:
                yield prefix + name, param
        for name, module in self._modules.items():
            if module is not None:
                yield from module.named_parameters(prefix + name + ".")

    def _move(self, method, **kw):
        for child in self._children():
            if child is not None and hasattr(child, method):
                getattr(child, method)(**kw)

    def to_cpu(self, non_blocking=False):
        self._move("to_cpu", non_blocking=non_blocking)

    def to_cuda(self, non_blocking=False):
        self._move("to_cuda", non_blocking=non_blocking)

    def to_cpu_async(self):
        self.to_cpu(non_blocking=True)

    def to_cuda_async(self):
        self.to_cuda(non_blocking=True)


class WeightModuleList(WeightModule):
    def __init__(self, modules=None):
        super().__init__()
        self._list = []
        for module in modules or []:
            self.append(module)

    def append(self, module):
        self.add_module(str(len(self._list)), module)
        self._list.append(module)

    def load(self, weight_dict):
        for m in self._list:
            m.load(weight_dict)

    def __getitem__(self, idx):
        return self._list[idx]

    def __len__(self):
        return len(self._list)

    def __iter__(self):
        return iter(self._list)
