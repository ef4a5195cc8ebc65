"""`randn_tensor` (torch_utils), BaseOutput, is_torch_version, logging stand-ins."""
import logging as _logging

import torch


class BaseOutput:
    """diffusers' BaseOutput is an OrderedDict-backed dataclass base; the reference only reads `.sample` attributes."""


def is_torch_version(op, version):
    from packaging import version as V

    cur, ref = V.parse(torch.__version__.split("+")[0]), V.parse(version)
    return {">=": cur >= ref, ">": cur > ref, "<": cur < ref, "<=": cur <= ref, "==": cur == ref}[op]


class logging:  # noqa: N801  (module-like namespace: `from diffusers.utils import logging; logging.get_logger(__name__)`)
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)
