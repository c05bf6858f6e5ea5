#!/bin/bash
for a in 6 18; do IDQN_CONV_PROF=9 timeout -k 10 120 python tools/probes/td_prof.py $a 2>&1 | grep "A ="; done
