class FromOriginalVAEMixin:
    pass
