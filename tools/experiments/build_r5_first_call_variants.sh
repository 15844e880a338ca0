#!/bin/bash
# Builds (here, on the CPU box: hipcc cross-compiles) the variant libraries tools/gpu_r5_first_call.sh times.  They start from HEAD.
# (The builds whose Y is garbage also skip everything behind mac_kernel -- r05_timing_only_on_top.patch -- or the exact fall-back
# would cost 13 s per step.)
set -e
cd "$(dirname "$0")/../.."
E=tools/experiments
python tools/build_variant.py sw            --patch $E/r05_mac_store_wave.patch
python tools/build_variant.py sw2           --patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_two_groups_on_top.patch
SW2="--patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_two_groups_on_top.patch"
python tools/build_variant.py sw2_q8        $SW2 --patch $E/r05_sw2_queue8_on_top.patch
python tools/build_variant.py sw2_q8_r4     $SW2 --patch $E/r05_sw2_queue8_rounds_of_four_on_top.patch
python tools/build_variant.py sw2_m3        $SW2 --patch $E/r05_sw2_drain3_on_top.patch
python tools/build_variant.py sw2_no_wait   $SW2 --patch $E/r05_sw2_no_wait_on_top.patch --patch $E/r05_timing_only_on_top.patch
python tools/build_variant.py iso_no_wait   --patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_iso_no_wait_on_top.patch --patch $E/r05_timing_only_on_top.patch
python tools/build_variant.py iso_no_wait_no_store --patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_iso_no_wait_no_store_on_top.patch --patch $E/r05_timing_only_on_top.patch
python tools/build_variant.py iso_no_consumer --patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_iso_no_consumer_on_top.patch --patch $E/r05_timing_only_on_top.patch
python tools/build_variant.py iso_w5        --patch $E/r05_iso_product_idle_fifth_wave.patch
python tools/build_variant.py iso_w5_2cu    --patch $E/r05_iso_product_idle_fifth_wave_two_per_cu.patch
python tools/build_variant.py no_y          --patch $E/r05_mac_ablation_no_y.patch
python tools/build_variant.py la2           --patch $E/r06_mac_class12_24_rows.patch
python tools/build_variant.py bound2        --patch $E/r06_bound_two_passes.patch
python tools/build_variant.py bound2_la2    --patch $E/r06_bound_two_passes.patch --patch $E/r06_mac_class12_24_rows.patch
python tools/build_variant.py probe_sw      --patch $E/r05_mac_store_wave.patch --patch $E/r05_store_wave_probe_on_top.patch
python tools/build_variant.py probe_product --patch $E/r05_product_probe.patch
python tools/build_variant.py probe_no_y    --patch $E/r05_mac_ablation_no_y.patch --patch $E/r05_product_probe.patch
ls -la sushi_amd/lib/
