#!/bin/bash
# TCC traffic of the scan kernel for a build variant: VARIANT="-DMK_SOMETHING=1" bash tools/pmc_traffic_variant.sh
# (scratch library via `make tuning`, selected with MK_LIBRARY; the shipped library is not touched)
cd $GRAFT_REPO_ROOT
make -s -C metakssd_amd/csrc tuning TUNING_OUT=/tmp/mk_variant_pmc VARIANT="$VARIANT" || exit 1
MK_TRAFFIC_VARIANT="${VARIANT:-tuning build}" MK_LIBRARY=/tmp/mk_variant_pmc/libmetakssd_hip.so bash tools/pmc_traffic.sh | tail -1
