#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  MF_DEVINGEST_TRACE=1 python tools/dmg_probe.py flip > /tmp/t.log 2>&1 &
  pid=$!
  sleep 8
  if kill -0 $pid 2>/dev/null; then
    echo "run $i hung (pid $pid)"; tail -2 /tmp/t.log
    if which gdb > /dev/null; then gdb -p $pid -batch -ex "thread apply all bt 12" 2>/dev/null | grep -v "^\[New\|^\[Thread\|warning" | cut -c1-200 | head -150; else echo "no gdb"; cat /proc/$pid/task/*/stack 2>/dev/null | head -50; for t in /proc/$pid/task/*; do echo "$t $(cat $t/comm) $(cat $t/wchan)"; done; fi
    kill -9 $pid; wait $pid 2>/dev/null
    break
  else echo "run $i ok"; fi
done
