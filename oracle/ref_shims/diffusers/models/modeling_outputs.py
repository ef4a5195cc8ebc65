from dataclasses import dataclass


@dataclass
class AutoencoderKLOutput:
    latent_dist: object = None
