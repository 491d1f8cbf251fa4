#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r2f
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python tools/replay.py --scans 50000 --progress 5000 > $out/replay_50k_lifelong.json 2> $out/replay_50k.err
tail -12 $out/replay_50k.err; cat $out/replay_50k_lifelong.json | cut -c1-2000
timeout 300 python tools/replay.py --scans 3000 --mode async --period 0.0005 > $out/replay_3k_async.json 2>> $out/replay_50k.err
cat $out/replay_3k_async.json | cut -c1-600
