#!/bin/bash
mkdir -p gpurun_out
PM355_PRE4=0 PM355_LIB=ab/ts.so timeout 300 python tools/seam_anatomy.py > gpurun_out/r3_c6_anatomy.txt 2>&1
tail -24 gpurun_out/r3_c6_anatomy.txt
