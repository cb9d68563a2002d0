#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout -k 10 200 python tools/diag_frame_kinematics.py 2000473 106 5 > $O/diag_kin_2000473.txt 2>&1; cat $O/diag_kin_2000473.txt | cut -c1-300 | tail -20
