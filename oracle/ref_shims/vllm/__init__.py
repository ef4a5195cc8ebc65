"""TEST INFRASTRUCTURE ONLY — stand-in for the `vllm` package (absent offline; the reference pins 0.9.2 in its benchmark doc) so that
the reference's fp8 operator classes (common/ops/mm/mm_weight.py:287-319) can be executed on CPU.  See _custom_ops.py."""
