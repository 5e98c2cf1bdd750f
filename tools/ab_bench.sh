#!/bin/bash
# Generic A/B of bench lines on one box: tools/ab_bench.sh <out-name> <reps> "<label>|<env assignments>|<bench args>" ...
# Each case runs `bench.py --no-cpu-baseline --no-extras <args>` with the assignments exported; cases alternate.
cd $GRAFT_REPO_ROOT
name=$1; reps=$2; shift 2
out=gpurun_out/$name.txt
: > $out
for rep in $(seq 1 $reps); do
  for c in "$@"; do
    label=${c%%|*}; rest=${c#*|}; envs=${rest%%|*}; args=${rest#*|}
    env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras $args > gpurun_out/ab_line.json 2> gpurun_out/ab_line.err || { echo "FAILED $label" >> $out; tail -5 gpurun_out/ab_line.err >> $out; cat $out; exit 1; }
    python -c "
import json; d = json.load(open('gpurun_out/ab_line.json'))
print('$label', d['value'], d['ms_per_step'], d['config'].get('replay_equals_eager'), d['config'].get('batches_per_replay'))" >> $out
  done
done
cat $out
