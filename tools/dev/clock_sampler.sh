#!/bin/bash
# dev tool: sample the card's clocks and power (rocm-smi, read-only) while a command runs; usage: clock_sampler.sh <out> <cmd...>
out=$1; shift
"$@" > "$out.cmd.log" 2>&1 &
pid=$!
: > "$out"
while kill -0 $pid 2>/dev/null; do
  { date +%s.%N; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power"; } >> "$out"
  sleep 0.05
done
wait $pid
