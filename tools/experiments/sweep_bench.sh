#!/bin/bash
cd "$(dirname "$0")/../.."
for v in none 1 0; do echo "ctx stream priority $v"; if [ $v = none ]; then unset DCA_CTX_STREAM_PRIORITY; else export DCA_CTX_STREAM_PRIORITY=$v; fi
  python tools/experiments/mf_twice.py 2>/dev/null | tail -2
  for cap in 448 512; do echo " cap $cap"; DCA_SWEEP_CAP=$cap python tools/experiments/mf_twice.py 2>/dev/null | tail -1; done
done
