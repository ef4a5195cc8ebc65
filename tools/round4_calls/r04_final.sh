#!/bin/bash
# Round 4, round-end batch: tools/final_run.sh without the 8-minute 50-step 14B end-to-end run.
RUN_TAG=r04_final E2E14=0 ATTN_AB_STEPS=10 bash tools/final_run.sh
