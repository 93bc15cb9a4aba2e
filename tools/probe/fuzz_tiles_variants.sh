for seed in 5056 5090 5112; do
  for v in "X=1" "SZ_TILE_PACK_INLINE=0" "SZ_MIGRATE_HOST=1" "SZ_TILE_INLINE=0" "SZ_CREC=0" "SZ_LEAN_NARROW=0"; do
    r=$(env $v SZ_PROBE_ANY_PATH=1 timeout -k 10 120 python tools/fuzz_tiles.py 1 $seed 2>&1 | grep "^case")
    echo "$seed $v :: $r"
  done
done
