#!/bin/bash
# tools/guarded.sh <seconds> <max resident GiB> <command ...>
# Runs the command under `timeout` with a watchdog on the resident memory of its whole process group: past the limit the group is killed (exit 137).
# For GPU-box runs of anything new: a runaway host allocation must end the program, not the machine (a lost box counts against the round).
T=$1; G=$2; shift; shift
setsid timeout -k 5 "$T" "$@" &
PID=$!
LIMIT_KB=$((G * 1024 * 1024))
while kill -0 $PID 2>/dev/null; do
    RSS=$(ps -o rss= -g $(ps -o sid= -p $PID 2>/dev/null | tr -d ' ') 2>/dev/null | awk '{s+=$1} END {print s+0}')
    if [ "${RSS:-0}" -gt "$LIMIT_KB" ]; then echo "guarded: resident memory ${RSS} KB over the ${G} GiB limit: killing"; kill -9 -- -$PID 2>/dev/null; kill -9 $PID 2>/dev/null; wait $PID 2>/dev/null; exit 137; fi
    sleep 0.2
done
wait $PID
