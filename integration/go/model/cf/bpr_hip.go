//go:build cgo && hip

package cf

// #include "gorse_hip.h"
import "C"

import (
	"context"
	"fmt"
	"time"

	"github.com/gorse-io/gorse/common/log"
	"github.com/gorse-io/gorse/common/monitor"
	"github.com/gorse-io/gorse/dataset"
	"github.com/samber/lo"
	"go.uber.org/zap"
)

// Fit the BPR model on one MI355X.  Everything around the epoch body -- Init draws, evaluation schedule, early stopping,
// logging, span, the returned Score -- is model.go:408-530 unchanged; the body (sampling + the SGD steps of one epoch,
// model.go:446-494) is one gorse_bpr_epoch / gorse_bpr_epoch_enqueue call.
func (bpr *BPR) Fit(ctx context.Context, trainSet, valSet dataset.CFSplit, config *FitConfig) Score {
	log.Logger().Info("fit bpr (hip)",
		zap.Int("train_set_size", trainSet.CountFeedback()),
		zap.Int("test_set_size", valSet.CountFeedback()),
		zap.Any("params", bpr.GetParams()),
		zap.Any("config", config))
	bpr.Init(trainSet)
	hm, err := newHipModel(ctx, &bpr.BaseMatrixFactorization, bpr.nFactors, trainSet, false)
	if err != nil {
		log.Logger().Error("fit bpr: no device", zap.Error(err))
		return Score{}
	}
	defer hm.close()
	// one sampler stream per Fit, seeded like the reference's worker generators (model.go:420-423)
	seed := uint64(bpr.GetRandomGenerator().Int63())
	mode := C.int32_t(C.GORSE_BPR_HOGWILD_STORES)
	if config.Jobs <= 1 {
		mode = C.GORSE_BPR_SEQUENTIAL // parallel.Parallel with one worker runs the samples strictly in order
	}
	const evalSeed = 0 // util.NewRandomGenerator(0) of dataset.go:244: the negatives' own stream
	evalStart := time.Now()
	score := hm.evaluateResident(valSet, trainSet, config.TopK, config.Candidates, evalSeed, NDCG, Precision, Recall)
	scores := []lo.Tuple2[int, float32]{{A: 0, B: score[0]}}
	log.Logger().Debug(fmt.Sprintf("fit bpr %v/%v", 0, bpr.nEpochs),
		zap.String("eval_time", time.Since(evalStart).String()),
		zap.Float32(fmt.Sprintf("NDCG@%v", config.TopK), score[0]),
		zap.Float32(fmt.Sprintf("Precision@%v", config.TopK), score[1]),
		zap.Float32(fmt.Sprintf("Recall@%v", config.TopK), score[2]))
	_, span := monitor.Start(ctx, "BPR.Fit", bpr.nEpochs)
	defer span.End()
	// enqueueDepth: epochs kept in flight between two evaluations (one running, one whose preparation runs under it).  The reference
	// checks ctx per sample (model.go:449); gorse_mf_epoch_throttle waits with the cancel flag in hand, so a cancelled context ends
	// the Fit within two epochs instead of at the next evaluation.  (host/gorse_cf.hpp kEnqueueDepth: the C++ twin, tested there.)
	const enqueueDepth = 2
	C.gorse_mf_epoch_times(hm.h, nil, nil, nil, 1) // a lent handle's earlier epochs do not count
	fitStart, fitEpochs := time.Now(), 0
	for epoch := 1; epoch <= bpr.nEpochs; epoch++ {
		// Between two evaluations the epochs are only ENQUEUED (gorse_bpr_epoch_enqueue): the sampler and the counting sort
		// of epoch e + 1 then run under the update kernel of epoch e (what bench.py times).  The epoch in front of an
		// evaluation -- and every epoch of the sequential schedule, which cannot be enqueued -- goes through the
		// synchronous call, which also polls the cancel flag.
		evalNext := epoch%config.Verbose == 0 || epoch == bpr.nEpochs
		var rc C.int32_t
		if !evalNext && mode != C.GORSE_BPR_SEQUENTIAL && ctx.Err() == nil {
			if rc = C.gorse_mf_epoch_throttle(hm.h, enqueueDepth, hm.cancel); rc == 0 {
				rc = C.gorse_bpr_epoch_enqueue(hm.h, C.int64_t(trainSet.CountFeedback()), C.float(bpr.lr), C.float(bpr.reg),
					C.uint64_t(seed), C.uint64_t(epoch), 0, mode)
			} else if rc == C.GORSE_ERR_CANCELLED {
				C.gorse_mf_synchronize(hm.h) // drain the (at most enqueueDepth) epochs in flight before the factors are pulled
			}
		} else {
			rc = C.gorse_bpr_epoch(hm.h, C.int64_t(trainSet.CountFeedback()), C.float(bpr.lr), C.float(bpr.reg),
				C.uint64_t(seed), C.uint64_t(epoch), 0, mode, hm.cancel, nil)
		}
		fitEpochs++
		if rc == C.GORSE_ERR_CANCELLED {
			log.Logger().Info("fit bpr canceled", zap.Int("epoch", epoch), zap.Error(ctx.Err()))
			hm.pull()
			return Score{}
		} else if rc != 0 {
			log.Logger().Error("fit bpr", zap.Error(hipError("gorse_bpr_epoch", rc)))
			return Score{}
		}
		if epoch%config.Verbose == 0 || epoch == bpr.nEpochs {
			// fit_time (model.go:496-503, what dashboards read): the mean DEVICE time of the epochs since the last evaluation
			// (hipEvents on the handle's update stream, gorse_mf_epoch_times) -- the host's clock around an enqueue measures the
			// enqueue.  Falls back to the host's mean over the period when the library timed another number of epochs.
			fitTime := time.Since(fitStart) / time.Duration(fitEpochs)
			var timed C.int64_t
			var devMs C.double
			if C.gorse_mf_epoch_times(hm.h, &timed, &devMs, nil, 1) == 0 && int(timed) == fitEpochs {
				fitTime = time.Duration(float64(devMs) / float64(timed) * float64(time.Millisecond))
			}
			evalStart = time.Now()
			score = hm.evaluateResident(valSet, trainSet, config.TopK, config.Candidates, evalSeed, NDCG, Precision, Recall)
			scores = append(scores, lo.Tuple2[int, float32]{A: epoch, B: score[0]})
			log.Logger().Info(fmt.Sprintf("fit bpr %v/%v", epoch, bpr.nEpochs),
				zap.String("fit_time", fitTime.String()),
				zap.String("eval_time", time.Since(evalStart).String()),
				zap.Float32(fmt.Sprintf("NDCG@%v", config.TopK), score[0]),
				zap.Float32(fmt.Sprintf("Precision@%v", config.TopK), score[1]),
				zap.Float32(fmt.Sprintf("Recall@%v", config.TopK), score[2]))
			if best, stop := earlyStop(scores, epoch, config.Patience); stop {
				log.Logger().Info("early stopping",
					zap.Int("best_epoch", best.A), zap.Float32("best_NDCG", best.B), zap.Int("patience", config.Patience))
				break
			}
			fitStart, fitEpochs = time.Now(), 0
		}
		span.Add(1)
	}
	hm.pull() // the [][]float32 rows alias the flat arrays: Marshal / GetUserFactor / Predict see the trained factors
	log.Logger().Info("fit bpr complete",
		zap.Float32(fmt.Sprintf("NDCG@%v", config.TopK), score[0]),
		zap.Float32(fmt.Sprintf("Precision@%v", config.TopK), score[1]),
		zap.Float32(fmt.Sprintf("Recall@%v", config.TopK), score[2]))
	return Score{NDCG: score[0], Precision: score[1], Recall: score[2]}
}
