#!/bin/bash
set -u
for mib in 16 64; do
  echo "BAZ_MUSIC_SINGLE_MIB=$mib"; BAZ_MUSIC_SINGLE_MIB=$mib python scripts/hostfed_extra.py 8192 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['runs'].items(): print('   %-45s %.3e items/s  %s items per call' % (k, v.get('items_per_s',0), v.get('items_per_work_call')))"
done
for lb in 512 2048; do
  echo "BAZ_MUSIC_INPUT_LOOKBACK=$lb (single 64 MiB)"; BAZ_MUSIC_INPUT_LOOKBACK=$lb python scripts/hostfed_extra.py 8192 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,v in d['runs'].items(): print('   %-45s %.3e items/s  %s items per call' % (k, v.get('items_per_s',0), v.get('items_per_work_call')))"
done
