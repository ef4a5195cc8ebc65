import torch.nn as nn


class ModelMixin(nn.Module):
    pass
