#!/bin/bash
O=gpurun_out/r04_widewalk; mkdir -p $O
python -m pytest tests/test_gpu_wide.py tests/test_gpu_save_load.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python -m pytest tests/test_gpu_beyond_4g.py -x -q -m gpu > $O/pytest_4g.txt 2>&1; tail -4 $O/pytest_4g.txt
python tests/test_gpu_beyond_4g.py dna > $O/beyond_4g_dna.json 2> $O/beyond_4g_dna.err; cat $O/beyond_4g_dna.json | head -c 1500
