//go:build cgo && hip

package floats

/*
#cgo LDFLAGS: -lgorse_hip
#include "gorse_hip.h"
*/
import "C"

import "unsafe"

// HIP marks the device GEMM in `feature` (the pattern of mm_openblas.go:21-23; add the constant next to OPENBLAS / MKL).
func init() {
	feature = feature | HIP
}

func cbool(v bool) C.int32_t {
	if v {
		return 1
	}
	return 0
}

// mm is floats.MM's backend (floats.go:241): row-major C = op(A) op(B) with the reference's own semantics -- the NN, TN
// and TT cases accumulate into C, NT overwrites it (mm.go:20-48) -- bit-exact to the AVX512 kernel.  Worth it for large
// products only: every call moves A, B and C over PCIe.
func mm(transA, transB bool, m, n, k int, a []float32, lda int, b []float32, ldb int, c []float32, ldc int) {
	if rc := C.gorse_hip_sgemm(0, cbool(transA), cbool(transB), C.int32_t(m), C.int32_t(n), C.int32_t(k),
		(*C.float)(unsafe.Pointer(&a[0])), C.int32_t(lda), (*C.float)(unsafe.Pointer(&b[0])), C.int32_t(ldb),
		(*C.float)(unsafe.Pointer(&c[0])), C.int32_t(ldc)); rc != 0 {
		panic("floats: " + C.GoString(C.gorse_hip_last_error())) // a length mismatch panics in the reference too
	}
}
