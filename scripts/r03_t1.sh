python scripts/ab.py "" "rows_per_tile=8" "rows_per_tile=4,tile_bits_dw=11" "rows_per_tile=2,tile_bits_dw=12" "rows_per_tile=2,tile_bits_dw=13,threads_dw=1024" 2>&1 | grep -v amdgpu.ids
