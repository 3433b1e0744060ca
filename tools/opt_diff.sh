#!/bin/bash
# GPU CLI vs CPU oracle on the lambda fixture for one option row; leaves both SAMs under gpurun_out/ (run on the GPU box)
# usage: tools/opt_diff.sh "<options>"
cd "$(dirname "$0")/.."
G=tests/golden/lambda
./salt_amd/bin/salt-idx -k 19 $G/genome.fa $G/snps.txt /tmp/optidx 2>/dev/null
case "$1" in *-p*) files="$G/reads_pe_1.fq $G/reads_pe_2.fq";; *) files="$G/reads_se.fq";; esac
./salt_amd/bin/salt $1 /tmp/optidx $files 2>gpurun_out/opt_gpu.err | grep -v '^@PG' > gpurun_out/opt_gpu.sam
./oracle/salt_oracle $1 /tmp/optidx $files 2>/dev/null | grep -v '^@PG' > gpurun_out/opt_ora.sam
cmp gpurun_out/opt_gpu.sam gpurun_out/opt_ora.sam && echo same
exit 0
