#!/usr/bin/env python
"""Checkpoint converter CLI — same flags as the reference's tools/convert/converter.py:640-707 for the parts of it that are on the
DiT path (quantisation to e4m3 / int8 with per-out-channel scales, per-block or chunked safetensors + index json).
Example (config #4's weights):
    python tools/convert_ckpt.py -s /models/Wan2.1-T2V-14B -o /models/Wan2.1-T2V-14B/fp8 -t wan_dit --quantized \
        --linear_dtype torch.float8_e4m3fn --non_linear_dtype torch.bfloat16 --save_by_block
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightx2v_amd import checkpoint as ck  # noqa: E402

DTYPES = {"torch.int8": torch.int8, "torch.float8_e4m3fn": torch.float8_e4m3fn, "torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16}


def main():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("-s", "--source", required=True)
    p.add_argument("-o", "--output", required=True)
    p.add_argument("-o_n", "--output_name", default="converted")
    p.add_argument("-c", "--chunk-size", type=int, default=100)
    p.add_argument("-t", "--model_type", choices=list(ck.MODEL_TYPE_KEYS), default="wan_dit")
    p.add_argument("-b", "--save_by_block", action="store_true")
    p.add_argument("--quantized", action="store_true")
    p.add_argument("--device", default="cpu")
    p.add_argument("--linear_dtype", choices=["torch.int8", "torch.float8_e4m3fn"], default="torch.float8_e4m3fn")
    p.add_argument("--non_linear_dtype", choices=["torch.float32", "torch.bfloat16", "torch.float16"], default="torch.float32")
    a = p.parse_args()
    if os.path.isfile(a.output):
        raise ValueError("Output path must be a directory, not a file")
    index = ck.convert_checkpoint(a.source, a.output, a.model_type, a.quantized, DTYPES[a.linear_dtype], DTYPES[a.non_linear_dtype], a.save_by_block, a.chunk_size,
                                  a.output_name, a.device)
    print(f"wrote {len(set(index['weight_map'].values()))} files, {index['metadata']['total_size'] / 2**20:.1f} MiB, to {a.output}")


if __name__ == "__main__":
    main()
