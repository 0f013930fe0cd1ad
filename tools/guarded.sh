#!/usr/bin/env bash
# guarded.sh <max RSS in GiB> <command ...>: runs the command and kills it when its resident set passes the limit
# (the GPU box kills the whole call at its host-memory cap; this keeps a runaway test from getting there).
lim_kb=$(( $1 * 1024 * 1024 )); shift
"$@" &
pid=$!
while kill -0 $pid 2>/dev/null; do
    rss=$(ps -o rss= -p $pid 2>/dev/null | tr -d ' ')
    if [ -n "$rss" ] && [ "$rss" -gt "$lim_kb" ]; then echo "guarded.sh: RSS $rss kB over the limit, killing $pid"; kill -9 $pid; fi
    sleep 1
done
wait $pid
