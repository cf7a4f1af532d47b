#!/bin/bash
# Host-side logic of libgecm under AddressSanitizer + UBSan (CPU only; no device code involved).
set -e
cd "$(dirname "$0")/.."
H=avx-ecm_amd/host
gcc -O1 -g -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -Wall -Wextra tools/host_sanitize.c $H/mpl.c $H/calc_lite.c $H/cunningham.c $H/gecm_plan.c $H/gecm_pair.c -lm -o /tmp/gecm_host_sanitize
ASAN_OPTIONS=detect_leaks=1 /tmp/gecm_host_sanitize
