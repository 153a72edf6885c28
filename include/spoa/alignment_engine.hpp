// spoa/alignment_engine.hpp — see spoa/spoa.hpp (the driver includes all three names, R/benchmarks/poa/msa_spoa_omp.cpp:20-22).
#include "spoa.hpp"
