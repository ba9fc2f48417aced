// Neighbourhood models (reference: similarity_base.hpp, itemcf.hpp, usercf.hpp) are OUT OF SCOPE of the
// MI355X hot path: the types exist so that apps/yelp compiles unchanged, and aborts with
// a clear message if one is selected at run time.
#ifndef CDAE_HOST_MODEL_RECSYS_SIMILARITY_BASE_HPP_
#define CDAE_HOST_MODEL_RECSYS_SIMILARITY_BASE_HPP_

#include <ostream>
#include <base/parallel.hpp>
#include <model/recsys/recsys_model_base.hpp>

namespace libcf {

enum SimilarityType { Jaccard, Cosine };
inline std::ostream& operator<<(std::ostream& o, const SimilarityType& st) { return o << (st == Jaccard ? "Jaccard" : "Cosine"); }

class SimilarityBase : public RecsysModelBase {
 public:
  SimilarityBase(size_t index_fg, size_t data_fg, SimilarityType st, size_t topk)
      : index_feature_group_(index_fg), data_feature_group_(data_fg), sim_type_(st), topk_(topk) {}
  void reset(const Data&) {
    LOG(FATAL) << "ItemCF / UserCF are not provided by this build: only the CDAE training hot path "
                  "(--method=CDAE) and the Popularity baseline are (SURVEY.md §2.1)";
  }
  void train_one_iteration(const Data&) {}
 protected:
  size_t index_feature_group_, data_feature_group_;
  SimilarityType sim_type_;
  size_t topk_;
};

}  // namespace libcf
#endif
