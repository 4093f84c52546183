#!/bin/bash
# GPU box helper: the weight-gradient kernel's in-kernel timeline (diagnostic build -DVMC_WGRAD_STAMPS:
# wall_clock64 stamps of every workgroup at loop start / loop end / stores drained / ticket / fold end).
# Compiles a stamped grad.o and links a stamped library UNDER A PATH OF ITS OWN (the product library in the
# tree is never touched: an interrupted run cannot leave the stamped build installed), selected for this
# one bench run through CGS_VMC_DIAGNOSTIC_LIBRARY.  Never quote run times of the stamped build.
set -uo pipefail
cd "$(dirname "$0")/.."
C=cgs_vmc_amd/csrc
D=$(mktemp -d /tmp/wgrad_stamps.XXXXXX)
trap 'rm -rf "$D"' EXIT
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DVMC_WGRAD_STAMPS -c $C/grad.hip -o "$D/grad_stamped.o" || exit 1
OBJS=$(ls $C/*.o | grep -v "/grad.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS "$D/grad_stamped.o" -o "$D/libcgsvmc_hip_stamped.so" || exit 1
CGS_VMC_DIAGNOSTIC_LIBRARY="$D/libcgsvmc_hip_stamped.so" CGS_VMC_ALLOW_EXTRA_BUILD=1 \
  python bench.py --steps 40 --warmup 5 --reps 1 --no-cpu-baseline --no-extra --no-timing "$@" 2>&1 | grep -v "^{" | tail -6 | cut -c1-400
