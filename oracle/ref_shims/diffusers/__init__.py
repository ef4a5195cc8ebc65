"""TEST INFRASTRUCTURE ONLY — stand-in for the `diffusers` package (absent offline), just enough for the reference's Hunyuan modules
to import and run on CPU:
  * schedulers/hunyuan/scheduler.py:3                      -> utils.torch_utils.randn_tensor
  * video_encoders/hf/autoencoder_kl_causal_3d/*.py        -> configuration_utils, loaders, utils(.accelerate_utils), models.*
Everything here is framework plumbing (config capture, base classes, an identity decorator) EXCEPT `models.attention_processor.Attention`,
which is arithmetic: it is restated from the published diffusers 0.29.2 class for the one configuration the reference instantiates
(UNetMidBlockCausal3D, unet_causal_3d_blocks.py:578-590).  Fixtures generated through this stub therefore pin the reference's own code
(convolutions, resnets, upsampling, tiling, blending, the mask) and leave that one op marked as restated."""
