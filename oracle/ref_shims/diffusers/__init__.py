"""Stand-in for the `diffusers` package (absent offline): only the symbol the reference's Hunyuan scheduler module
imports at load time (schedulers/hunyuan/scheduler.py:3).  Test infrastructure only."""
