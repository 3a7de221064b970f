// dvo_slam/constraints/constraint_proposal_validator.h -- staged validation of loop-closure proposals.
//
// Interface and decision logic of the reference's ConstraintProposalValidator
// (dvo_slam/include/dvo_slam/constraints/constraint_proposal_validator.h:36-80,
// src/constraints/constraint_proposal_validator.cpp:30-165): every stage re-tracks the surviving proposals with its own
// tracker configuration (typically a coarse level-3-only screening, then a 3->1 refinement), lets its voters decide,
// drops the rejected ones and hands the inverse of each tracked transform to the next stage as initial guess.
//
// What differs is where the time goes: the reference tracks the proposals of a stage one after another (and shards the
// proposal list over a TBB pool, keyframe_graph.cpp:524-593).  Here ALL proposals of a stage -- including the
// cross-validation twins -- are aligned in ONE device batch (DenseTracker::matchBatch -> dvo_hip_match_batch), which is
// the workload the batched kernels are built for.  The proposal list is therefore validated whole; results are
// independent of how a caller might have split it.
#pragma once

#include <algorithm>
#include <iostream>
#include <memory>
#include <vector>

#include "dvo/dense_tracking.h"
#include "dvo_slam/constraints/constraint_proposal.h"
#include "dvo_slam/constraints/constraint_proposal_voter.h"

namespace dvo_slam {
namespace constraints {

struct ConstraintProposalValidator {
 public:
  struct Stage {
   private:
    friend struct ConstraintProposalValidator;
    int Id;
    bool OnlyKeepBest;
    dvo::DenseTracker::Config TrackingConfig;
    ConstraintProposalVoterVector Voters;
    explicit Stage(int id) : Id(id), OnlyKeepBest(false) {}

   public:
    Stage& keepBest() { OnlyKeepBest = true; return *this; }
    Stage& keepAll() { OnlyKeepBest = false; return *this; }
    Stage& trackingConfig(const dvo::DenseTracker::Config& cfg) { TrackingConfig = cfg; return *this; }
    Stage& addVoter(ConstraintProposalVoter* v) { Voters.push_back(ConstraintProposalVoterPtr(v)); return *this; }
    int id() const { return Id; }
    const dvo::DenseTracker::Config& trackingConfig() const { return TrackingConfig; }
  };
  typedef std::vector<Stage> StageVector;

  ConstraintProposalValidator() {}
  virtual ~ConstraintProposalValidator() {}

  Stage& createStage(int id) {
    stages_.push_back(Stage(id));
    return stages_.back();
  }

  void validate(ConstraintProposalVector& proposals, bool debug = false) {
    for (StageVector::iterator stage = stages_.begin(); stage != stages_.end(); ++stage) {
      for (size_t i = 0; i < proposals.size(); ++i) {
        proposals[i]->clearVotes();
        proposals[i]->TrackingResult.clearStatistics();
      }
      validate(*stage, proposals, debug);
      if (debug) {
        std::cout << "Stage " << stage->Id << ":" << std::endl;
        for (size_t i = 0; i < proposals.size(); ++i) proposals[i]->printVotingResults(std::cout, "  ");
      }
      proposals.erase(std::remove_if(proposals.begin(), proposals.end(), [](const ConstraintProposalPtr& p) { return p->Reject(); }),
                      proposals.end());
      if (stage->OnlyKeepBest) keepBest(proposals);
      // the next stage starts from this stage's estimate
      for (size_t i = 0; i < proposals.size(); ++i) proposals[i]->InitialTransformation = proposals[i]->TrackingResult.Transformation.inverse();
    }
  }

  // of several proposals linking the same two keyframes (in either direction) only the highest-scoring one stays, at
  // the position of the first
  void keepBest(ConstraintProposalVector& proposals) {
    for (size_t i = 0; i < proposals.size(); ++i)
      for (size_t j = i + 1; j < proposals.size();) {
        if (!proposals[i]->isConstraintBetweenSameFrames(*proposals[j])) {
          ++j;
          continue;
        }
        if (proposals[j]->TotalScore() > proposals[i]->TotalScore()) proposals[i].swap(proposals[j]);
        proposals.erase(proposals.begin() + long(j));
      }
  }

 protected:
  // Tracking results of all proposals of a stage.  One device batch; a subclass may substitute another source
  // (tests drive the decision logic with tabulated results).
  virtual void track(const dvo::DenseTracker::Config& cfg, ConstraintProposalVector& proposals) {
    tracker_.configure(cfg);
    std::vector<dvo::core::RgbdImagePyramid*> refs, curs;
    std::vector<dvo::DenseTracker::Result*> results;
    for (size_t i = 0; i < proposals.size(); ++i) {
      ConstraintProposal& p = *proposals[i];
      p.TrackingResult.Transformation = p.InitialTransformation;
      refs.push_back(p.Reference->image().get());
      curs.push_back(p.Current->image().get());
      results.push_back(&p.TrackingResult);
    }
    tracker_.matchBatch(refs, curs, results);
  }

 private:
  StageVector stages_;
  dvo::DenseTracker tracker_;

  void validate(Stage& stage, ConstraintProposalVector& proposals, bool debug) {
    for (size_t v = 0; v < stage.Voters.size(); ++v) stage.Voters[v]->createAdditionalProposals(proposals);
    track(stage.TrackingConfig, proposals);
    for (size_t i = 0; i < proposals.size(); ++i) {
      ConstraintProposal& p = *proposals[i];
      for (size_t v = 0; v < stage.Voters.size(); ++v) {
        p.Votes.push_back(stage.Voters[v]->vote(p, debug));
        if (p.Votes.back().Decision == ConstraintProposal::Vote::Reject && !debug) break;   // the first veto settles it
      }
    }
    for (size_t v = stage.Voters.size(); v-- > 0;) stage.Voters[v]->removeAdditionalProposals(proposals);
  }
};

typedef std::shared_ptr<ConstraintProposalValidator> ConstraintProposalValidatorPtr;

}  // namespace constraints
}  // namespace dvo_slam
