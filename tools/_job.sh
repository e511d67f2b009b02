#!/bin/bash
bash tools/prof_r06.sh 2>&1 | grep -v "^{" | grep "driver command\|world 8\|wrote" 
bash tools/gpu_suite.sh 2>&1 | tail -2
