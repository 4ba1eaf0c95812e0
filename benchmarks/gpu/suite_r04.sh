#!/bin/bash
# the whole -m gpu suite; output under gpurun_out/r04_suite/
O=gpurun_out/r04_suite; mkdir -p $O
( time python -m pytest tests -q -m gpu ) > $O/pytest_gpu.txt 2>&1; tail -8 $O/pytest_gpu.txt
