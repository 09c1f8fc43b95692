for e in 134217728 268435456 402653184; do python tools/workload.py oct_img 1e8 peel_events=$e 2>&1 | grep "^oct" | tail -1 | sed "s/^/events=$e /"; done
