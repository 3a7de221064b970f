// dvo_slam/constraints/constraint_proposal_voter.h -- the acceptance tests applied to a tracked loop-closure proposal
// (reference: dvo_slam/include/dvo_slam/constraints/constraint_proposal_voter.h:37-110,
// src/constraints/constraint_proposal_voter.cpp:30-186).  Pure host logic over DenseTracker::Result.
#pragma once

#include <cassert>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <sstream>
#include <utility>
#include <vector>

#include "dvo_slam/constraints/constraint_proposal.h"

namespace dvo_slam {
namespace constraints {

struct ConstraintProposalVoter {
  virtual ~ConstraintProposalVoter() {}
  // voters may ask for tracking results of extra proposals and must take them out again afterwards
  virtual void createAdditionalProposals(ConstraintProposalVector&) {}
  virtual void removeAdditionalProposals(ConstraintProposalVector&) {}
  // must set the decision; the reason only on request
  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) = 0;
};
typedef std::shared_ptr<ConstraintProposalVoter> ConstraintProposalVoterPtr;
typedef std::vector<ConstraintProposalVoterPtr> ConstraintProposalVoterVector;

namespace detail {
inline ConstraintProposal::Vote decide(bool accept, double score, bool provide_reason, const char* name, double lhs, const char* op, double rhs) {
  ConstraintProposal::Vote v;
  v.Decision = accept ? ConstraintProposal::Vote::Accept : ConstraintProposal::Vote::Reject;
  v.Score = score;
  if (provide_reason) {
    std::stringstream reason;
    reason << name << " " << lhs;
    if (op) reason << " " << op << " " << rhs;
    v.Reason = reason.str();
  }
  return v;
}
}  // namespace detail

// Tracks every proposal in both directions; accepts when the two results compose to (almost) the identity.
struct CrossValidationVoter : public ConstraintProposalVoter {
  double TranslationThreshold;
  explicit CrossValidationVoter(double threshold) : TranslationThreshold(threshold) {}
  CrossValidationVoter(const CrossValidationVoter& o) : ConstraintProposalVoter(), TranslationThreshold(o.TranslationThreshold) {}

  virtual void createAdditionalProposals(ConstraintProposalVector& proposals) {
    const size_t n = proposals.size();
    for (size_t i = 0; i < n; ++i) {
      ConstraintProposalPtr twin = proposals[i]->createInverseProposal();
      twins_.push_back(std::make_pair(proposals[i].get(), twin.get()));
      proposals.push_back(twin);
    }
  }

  // of each (proposal, twin) pair only the better one stays: the original if it is accepted and scores at least as high
  virtual void removeAdditionalProposals(ConstraintProposalVector& proposals) {
    for (size_t k = 0; k < twins_.size(); ++k) {
      ConstraintProposal* first = twins_[k].first;
      ConstraintProposal* second = twins_[k].second;
      const ConstraintProposal* loser = (first->TotalScore() >= second->TotalScore() && first->Accept()) ? second : first;
      for (ConstraintProposalVector::iterator it = proposals.begin(); it != proposals.end(); ++it)
        if (it->get() == loser) {
          proposals.erase(it);
          break;
        }
    }
    twins_.clear();
  }

  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) {
    const ConstraintProposal* twin = 0;
    for (size_t k = 0; k < twins_.size() && !twin; ++k) {
      if (twins_[k].first == &proposal) twin = twins_[k].second;
      else if (twins_[k].second == &proposal) twin = twins_[k].first;
    }
    assert(twin != 0);
    double m[16];
    dvo::compat::affine_to_rowmajor(twin->TrackingResult.Transformation * proposal.TrackingResult.Transformation, m);
    const double norm = std::sqrt(m[3] * m[3] + m[7] * m[7] + m[11] * m[11]);
    return detail::decide(norm <= TranslationThreshold, 0.0, provide_reason, "CrossValidation", norm, "<=", TranslationThreshold);
  }

 private:
  std::vector<std::pair<ConstraintProposal*, ConstraintProposal*> > twins_;
};

// quality of the result relative to the reference keyframe's own odometry baseline (entropy ratio by default)
struct TrackingResultEvaluationVoter : public ConstraintProposalVoter {
  double RatioThreshold;
  explicit TrackingResultEvaluationVoter(double threshold) : RatioThreshold(threshold) {}
  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) {
    const double ratio = proposal.Reference->evaluation()->ratioWithAverage(proposal.TrackingResult);
    return detail::decide(ratio >= RatioThreshold, ratio, provide_reason, "TrackingResultValidation", ratio, ">=", RatioThreshold);
  }
};

// fraction of the selected reference pixels that still produced a constraint in the last accepted iteration
struct ConstraintRatioVoter : public ConstraintProposalVoter {
  double RatioThreshold;
  explicit ConstraintRatioVoter(double threshold) : RatioThreshold(threshold) {}
  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) {
    const dvo::DenseTracker::LevelStats& l = proposal.TrackingResult.Statistics.Levels.back();
    const double ratio = l.HasIterationWithIncrement() ? double(l.LastIterationWithIncrement().ValidConstraints) / double(l.ValidPixels) : 0.0;
    return detail::decide(ratio >= RatioThreshold, 0.0, provide_reason, "ConstraintRatio", ratio, ">=", RatioThreshold);
  }
};

struct NaNResultVoter : public ConstraintProposalVoter {
  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) {
    const bool nan = proposal.TrackingResult.isNaN();
    return detail::decide(!nan, 0.0, provide_reason, "NaNResult", double(nan), 0, 0.0);
  }
};

// neighbours in the keyframe chain are already linked by odometry
struct OdometryConstraintVoter : public ConstraintProposalVoter {
  virtual ConstraintProposal::Vote vote(const ConstraintProposal& proposal, bool provide_reason) {
    const bool odometry = std::abs(int(proposal.Reference->id()) - int(proposal.Current->id())) <= 1;
    return detail::decide(!odometry, 0.0, provide_reason, "OdometryConstraint", double(odometry), 0, 0.0);
  }
};

}  // namespace constraints
}  // namespace dvo_slam
