#!/bin/bash
for st in 0 4 8 16; do echo "== stagger $st"; timeout -k 10 120 tools/bin/ubench_ffn2_st$st 2>&1 | head -3; done
