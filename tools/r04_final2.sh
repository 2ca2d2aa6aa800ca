#!/bin/bash
# second final round of round 4: the whole GPU suite, then the profile rounds profiles/r04_* are promoted from
mkdir -p gpurun_out
timeout 1300 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/final_round_r04.sh
