#!/bin/bash
out=gpurun_out/s2b; mkdir -p $out
timeout 300 python tools/exp_pipe.py > $out/pipe.txt 2>&1
tail -20 $out/pipe.txt
