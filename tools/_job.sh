#!/bin/bash
bash tools/prof_r06.sh 2>&1 | tail -30
bash tools/gpu_suite.sh 2>&1 | tail -4
