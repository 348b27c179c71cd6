//go:build !cgo || !hip

package logics

import "github.com/gorse-io/gorse/common/ann"

// the reference's index behind the field type logics/cf.go now uses (see cf_hip.go)
func newItemsIndex() itemsIndex { return ann.NewHNSW(distance) }
