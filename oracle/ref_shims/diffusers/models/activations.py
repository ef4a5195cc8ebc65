import torch.nn as nn


def get_activation(act_fn):
    table = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}
    return table[act_fn.lower()]()
