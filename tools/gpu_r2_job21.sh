#!/bin/bash
R=$PWD; O=$R/gpurun_out; mkdir -p $O
{ timeout 600 python tools/exp_tail.py int8; timeout 600 python tools/exp_tail.py f32; } > $O/r2_tail_round.txt 2>&1
cat $O/r2_tail_round.txt
