// Inverse-index (CSR) builder shared by the backward ops (grad.hip) and the spatially
// sorted neighbour searches (nn_sorted.hip): integer-atomic histogram, per-cloud scan,
// fill, optional ascending sort inside each row.  Defined in grad.hip.
#pragma once
#include "common.h"

namespace prcnn {

struct CsrWs {
    int *off;        // (b, nd+1): after the build, off[bi][k] = first slot of row k
    int *cursor;     // (b, nd)   scratch
    int *list;       // (b, E)    entry numbers grouped by row
};

size_t csr_ws_ints(long long b, long long nd, long long E);
CsrWs csr_carve(void *ws, long long b, long long nd, long long E);
int csr_build(int b, int nd, long long E, const int *keys, CsrWs w, bool sort_rows, hipStream_t st);

}  // namespace prcnn
