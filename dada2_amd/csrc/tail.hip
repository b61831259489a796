// tail.hip — the persistent round tail (rounds3.inc.hip: k3_tail) as a translation unit of its own.
//
// It runs the phase bodies of rounds2.inc.hip (shuffle_body, pupdate_body, birth_body - the same source the launch chains of
// kernels.hip compile) inside one loop that spans every phase of a round.  Compiled with the default pipeline, everything cheap
// and invariant in that loop - lane-derived LDS addresses, comparisons of the thread index with constants, the fp64 polynomial
// coefficients of the Poisson tail - was hoisted out of it and then SPILLED across it: 128 VGPRs + 97 spilled + 448 bytes of
// scratch per lane, 114 scratch loads inside phases that are chains of dependent memory round trips.  The Makefile
// builds this file with -mllvm -disable-machine-licm (the other kernels keep the default: their inner loops want the hoisting):
// 43 spilled VGPRs, 272 bytes.
// Same box, back to back (profiles/r07b_tail_codegen_variants.jsonl): k3_tail 76.5 -> 70.5 ms per pass at 10^6 uniques.  Reading the
// argument block through a pointer instead of taking it by value (which halves the spilled SGPRs) was measured too and is NOT
// done: every phase then starts with a chain of scalar loads that miss the scalar cache behind the barrier's invalidate (+12 ms).
#include <algorithm>
#include <cstdint>

#include "engine.h"
#include "gcn.h"
#include "knobs.h"
#include "ppois.h"
#include "rounds_common.h"

#define D2_TAIL_TU 1   // rounds2.inc.hip: device bodies only (its kernels and launch wrappers belong to kernels.hip)

namespace d2 {

#include "rounds2.inc.hip"
#include "rounds3.inc.hip"

}  // namespace d2
