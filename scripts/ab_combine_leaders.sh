for L in 1 2 3; do echo "== GBX_COMBINE_LEADERS=$L"; GBX_COMBINE_LEADERS=$L python scripts/refdrivers_large.py bsw phmm 2>&1 | grep -E "combine=1" | grep -E "\-t (64|16) " ; done
