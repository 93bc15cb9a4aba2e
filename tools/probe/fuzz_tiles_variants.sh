#!/bin/bash
# a fuzz case (tools/fuzz_tiles.py) under the feature switches:  tools/probe/fuzz_tiles_variants.sh "<seed> [mixed|walls]" ...
for a in "$@"; do
  for v in "X=1" "SZ_MIGRATE_HOST=1" "SZ_TILE_INLINE=0" "SZ_CREC=0" "SZ_LEAN_NARROW=0" "SZ_ELEMS_RIDE=0" "SZ_TILE_FORCING_TAIL=1"; do
    r=$(env $v SZ_PROBE_ANY_PATH=1 timeout -k 10 120 python tools/fuzz_tiles.py 1 $a 2>&1 | grep "^case" | cut -c1-220)
    echo "$a $v :: $r"
  done
done
