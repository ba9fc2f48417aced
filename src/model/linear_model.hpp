// LinearModel — generic regressor of the reference (src/model/linear_model.hpp), included by apps/yelp but
// never instantiated there.  OUT OF SCOPE: declaration only.
#ifndef CDAE_HOST_MODEL_LINEAR_MODEL_HPP_
#define CDAE_HOST_MODEL_LINEAR_MODEL_HPP_
#include <model/model_base.hpp>
namespace libcf {
struct LinearModelConfig { double lambda = 0.; LossType lt = SQUARE; PenaltyType pt = L2; bool using_bias_term = true; };
class LinearModel : public ModelBase, public SGDBase {
 public:
  LinearModel() = default;
  explicit LinearModel(const LinearModelConfig&) {}
};
}  // namespace libcf
#endif
