// libcf::BPR (pairwise ranking) on MI355X: the reference's class (src/model/recsys/bpr.hpp:12-106) — an IMF whose training loop
// takes (positive, sampled negative) pairs (bpr.hpp:56-106) — forwarding to the same C ABI handle with pairwise = 1
// (cdae_hip_create_mf).  Selected by apps/yelp with --method=BPR (yelp.cpp:144-165).
#ifndef CDAE_HOST_MODEL_RECSYS_BPR_HPP_
#define CDAE_HOST_MODEL_RECSYS_BPR_HPP_

#include <model/recsys/imf.hpp>

namespace libcf {

struct BPRConfig {
  BPRConfig() = default;
  double learn_rate = 0.1;
  double beta = 1.;
  double lambda = 0.01;
  LossType lt = LOG;
  PenaltyType pt = L2;
  size_t num_dim = 10;
  size_t num_neg = 5;
  bool using_bias_term = true;
  bool using_adagrad = true;
};

class BPR : public IMF {
 public:
  explicit BPR(const BPRConfig& mcfg) {
    IMFConfig c;
    c.learn_rate = mcfg.learn_rate; c.beta = mcfg.beta; c.lambda = mcfg.lambda; c.lt = mcfg.lt; c.pt = mcfg.pt;
    c.num_dim = mcfg.num_dim; c.num_neg = mcfg.num_neg; c.using_bias_term = mcfg.using_bias_term; c.using_adagrad = mcfg.using_adagrad;
    configure(c, true, "BPR");
  }
  void reset(const Data& data_set) { IMF::reset(data_set); }        // bpr.hpp:52-54
};

}  // namespace libcf
#endif
