//go:build cgo && hip

// worker/pipeline.go:403-448 for many users at once: with database.vector = "hip://" the collaborative recommendations of a
// batch of users are ONE QueryVectorsBatch on the device (exact -dot top-k with the hidden-item mask applied inside the
// search) instead of one QueryVectors round trip per user.  The master needs no change at all: its publishing step
// (master/tasks.go:930-962) talks to vectors.Database, and "hip://" is one (storage/vectors/hip.go registers the prefix).
// Not compiled here: no Go toolchain in the build image.
package worker

import (
	"context"
	"time"

	mapset "github.com/deckarep/golang-set/v2"
	"github.com/gorse-io/gorse/common/log"
	"github.com/gorse-io/gorse/storage/cache"
	"github.com/gorse-io/gorse/storage/vectors"
	"github.com/pkg/errors"
	"go.uber.org/zap"
)

// bulkQuerier is what storage/vectors/hip.go offers beyond vectors.Database.
type bulkQuerier interface {
	QueryVectorsBatch(ctx context.Context, collection string, queries [][]float32, categories []string, topK int) ([][]vectors.ScoredVector, error)
}

// updateCollaborativeRecommendBulk is updateCollaborativeRecommend for a batch of users.  topK is the largest
// CacheSize + |excludeSet| of the batch: every user gets at least what the per-user call would have fetched.
func (p *Pipeline) updateCollaborativeRecommendBulk(ctx context.Context, matrixFactorizationID int64, userIDs []string,
	userEmbeddings [][]float32, excludeSets []mapset.Set[string]) error {
	bulk, ok := p.VectorClient.(bulkQuerier)
	if !ok { // not the hip:// backend: the reference's loop
		for t := range userIDs {
			if err := p.updateCollaborativeRecommend(ctx, matrixFactorizationID, userIDs[t], userEmbeddings[t], excludeSets[t]); err != nil {
				return err
			}
		}
		return nil
	}
	topK := 0
	for _, s := range excludeSets {
		topK = max(topK, p.Config.Recommend.CacheSize+s.Cardinality())
	}
	localStartTime := time.Now()
	results, err := bulk.QueryVectorsBatch(ctx, vectors.CollaborativeFilteringCollection(matrixFactorizationID), userEmbeddings, nil, topK)
	if err != nil {
		return errors.WithStack(err)
	}
	for t, scoredVectors := range results {
		want := p.Config.Recommend.CacheSize + excludeSets[t].Cardinality()
		recommend := make([]cache.Score, 0, len(scoredVectors))
		for e, vector := range scoredVectors {
			if e >= want {
				break
			}
			if !excludeSets[t].Contains(vector.Id) {
				recommend = append(recommend, cache.Score{Id: vector.Id, Score: float64(vector.Score), Categories: vector.Categories, Timestamp: localStartTime})
			}
		}
		if err := p.CacheClient.AddScores(ctx, cache.CollaborativeFiltering, userIDs[t], recommend); err != nil {
			log.Logger().Error("failed to cache collaborative filtering recommendation result", zap.String("user_id", userIDs[t]), zap.Error(err))
			return errors.WithStack(err)
		}
		if err := p.CacheClient.Set(ctx,
			cache.Time(cache.Key(cache.CollaborativeFilteringUpdateTime, userIDs[t]), localStartTime),
			cache.String(cache.Key(cache.CollaborativeFilteringDigest, userIDs[t]), p.Config.Recommend.Collaborative.Hash(&p.Config.Recommend)),
		); err != nil {
			return errors.WithStack(err)
		}
		if err := p.CacheClient.DeleteScores(ctx, []string{cache.CollaborativeFiltering}, cache.ScoreCondition{Before: &localStartTime, Subset: new(userIDs[t])}); err != nil {
			return errors.WithStack(err)
		}
	}
	return nil
}
