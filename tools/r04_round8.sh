#!/bin/bash
set -u
R=$PWD; O=$R/gpurun_out/${1:-r04l}; mkdir -p $O
timeout 600 python -m pytest tests/test_detect.py tests/test_cli.py -q -m gpu > $O/tests_f4.log 2>&1; tail -30 $O/tests_f4.log
timeout 1800 python -m pytest tests -q -m gpu --deselect tests/test_detect.py --deselect tests/test_cli.py > $O/tests.log 2>&1; tail -6 $O/tests.log
