// poa_quad_defs.h -- constants shared by poa_quad_core.h and its back ends
#pragma once
#define PQ_NEG (-0x20000000)
#define PQ_TGB 0x20000000
#define PQ_O1 4
#define PQ_E1 2
#define PQ_O2 24
#define PQ_E2 1
#define PQ_MATCH 2
#define PQ_MISMATCH 4
#define PQ_COL_SINK 0x7FFFFFFE
#define PQ_COL_NEW 0x7FFFFFFF
#define PQ_RING 4     // LDS ring rows per group; a predecessor 1..3 rows back is read from the ring
#define PQ_GD 4       // "no path" guard cells on each side of a ring row
#define PQ_INT_MIN (-0x7fffffff - 1)

