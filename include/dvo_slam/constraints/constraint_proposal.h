// dvo_slam/constraints/constraint_proposal.h -- one loop-closure hypothesis: "keyframe Current is visible from keyframe
// Reference under InitialTransformation", its tracking result and the voters' verdicts
// (reference: dvo_slam/include/dvo_slam/constraints/constraint_proposal.h:37-88, src/constraints/constraint_proposal.cpp:30-112).
#pragma once

#include <cstdlib>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#include "dvo/dense_tracking.h"
#include "dvo_slam/keyframe.h"

namespace dvo_slam {
namespace constraints {

struct ConstraintProposal {
  struct Vote {
    enum Enum { Accept, Reject };
    Enum Decision;        // hard decision
    double Score;         // ranks competing proposals between the same two frames
    std::string Reason;   // filled on request
    Vote() : Decision(Reject), Score(0.0) {}
  };
  typedef std::vector<Vote> VoteVector;
  typedef std::shared_ptr<ConstraintProposal> Ptr;

  KeyframePtr Reference, Current;
  dvo::core::AffineTransformd InitialTransformation;
  dvo::DenseTracker::Result TrackingResult;
  VoteVector Votes;

  ConstraintProposal() { InitialTransformation.setIdentity(); }

  // no prior on the relative pose
  static Ptr createWithIdentity(const KeyframePtr& reference, const KeyframePtr& current) {
    Ptr p(new ConstraintProposal());
    p->Reference = reference;
    p->Current = current;
    return p;
  }
  // relative pose taken from the current map estimate
  static Ptr createWithRelative(const KeyframePtr& reference, const KeyframePtr& current) {
    Ptr p = createWithIdentity(reference, current);
    p->InitialTransformation = current->pose().inverse() * reference->pose();
    return p;
  }
  // the same hypothesis with the roles of the two frames exchanged
  Ptr createInverseProposal() const {
    Ptr p(new ConstraintProposal());
    p->Reference = Current;
    p->Current = Reference;
    p->InitialTransformation = InitialTransformation.inverse();
    return p;
  }

  double TotalScore() const {
    double s = 0.0;
    for (size_t i = 0; i < Votes.size(); ++i) s += Votes[i].Score;
    return s;
  }
  bool Reject() const {
    for (size_t i = 0; i < Votes.size(); ++i)
      if (Votes[i].Decision == Vote::Reject) return true;
    return false;
  }
  bool Accept() const { return !Reject(); }
  void clearVotes() { Votes.clear(); }

  bool isConstraintBetweenSameFrames(const ConstraintProposal& o) const {
    const int r = Reference->id(), c = Current->id(), orf = o.Reference->id(), oc = o.Current->id();
    return (r == orf && c == oc) || (r == oc && c == orf);
  }

  void printVotingResults(std::ostream& out, const std::string& indent = "") const {
    out << indent << "Proposal " << Reference->id() << "->" << Current->id() << " " << (Accept() ? "accept" : "reject") << std::endl;
    for (size_t i = 0; i < Votes.size(); ++i) out << indent << "  " << Votes[i].Reason << std::endl;
  }
};

typedef ConstraintProposal::Ptr ConstraintProposalPtr;
typedef std::vector<ConstraintProposalPtr> ConstraintProposalVector;

}  // namespace constraints
}  // namespace dvo_slam
