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

// Fit the ALS (eALS) model on one MI355X: model.go:609-775 with the epoch body (S = sum q q^T, user sweep, S = sum p p^T,
// item sweep; model.go:641-738) as one gorse_als_epoch call.
func (als *ALS) Fit(ctx context.Context, trainSet, valSet dataset.CFSplit, config *FitConfig) Score {
	log.Logger().Info("fit als (hip)",
		zap.Int("train_set_size", trainSet.CountFeedback()),
		zap.Int("test_set_size", valSet.CountFeedback()),
		zap.Any("params", als.GetParams()),
		zap.Any("config", config))
	als.Init(trainSet)
	hm, err := newHipModel(ctx, &als.BaseMatrixFactorization, als.nFactors, trainSet, true)
	if err != nil {
		log.Logger().Error("fit als: no device", zap.Error(err))
		return Score{}
	}
	defer hm.close()
	const evalSeed = 0 // util.NewRandomGenerator(0) of dataset.go:244: the negatives' own stream
	score := hm.evaluateResident(valSet, trainSet, config.TopK, config.Candidates, evalSeed, NDCG, Precision, Recall)
	scores := []lo.Tuple2[int, float32]{{A: 0, B: score[0]}}
	_, span := monitor.Start(ctx, "ALS.Fit", als.nEpochs)
	defer span.End()
	for epoch := 1; epoch <= als.nEpochs; epoch++ {
		fitStart := time.Now()
		rc := C.gorse_als_epoch(hm.h, C.float(als.weight), C.float(als.reg), hm.cancel)
		if rc == C.GORSE_ERR_CANCELLED {
			log.Logger().Info("fit als canceled", zap.Int("epoch", epoch), zap.Error(ctx.Err()))
			hm.pull()
			return Score{}
		} else if rc != 0 {
			log.Logger().Error("fit als", zap.Error(hipError("gorse_als_epoch", rc)))
			return Score{}
		}
		fitTime := time.Since(fitStart)
		if epoch%config.Verbose == 0 || epoch == als.nEpochs {
			evalStart := time.Now()
			score = hm.evaluateResident(valSet, trainSet, config.TopK, config.Candidates, evalSeed, NDCG, Precision, Recall)
			scores = append(scores, lo.Tuple2[int, float32]{A: epoch, B: score[0]})
			log.Logger().Info(fmt.Sprintf("fit als %v/%v", epoch, als.nEpochs),
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
		}
		span.Add(1)
	}
	hm.pull()
	log.Logger().Info("fit als complete",
		zap.Float32(fmt.Sprintf("NDCG@%v", config.TopK), score[0]),
		zap.Float32(fmt.Sprintf("Precision@%v", config.TopK), score[1]),
		zap.Float32(fmt.Sprintf("Recall@%v", config.TopK), score[2]))
	return Score{NDCG: score[0], Precision: score[1], Recall: score[2]}
}
