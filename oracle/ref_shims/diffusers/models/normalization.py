class AdaGroupNorm:
    def __init__(self, *a, **k):
        raise NotImplementedError("AdaGroupNorm is not used by the Hunyuan VAE decode path (time_embedding_norm='default')")


class RMSNorm:
    def __init__(self, *a, **k):
        raise NotImplementedError("diffusers RMSNorm is not used by the Hunyuan VAE decode path")
