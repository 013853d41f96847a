// ORACLE — TEST INFRASTRUCTURE ONLY (see linalg.h header).  PARITY UNPINNED: the reference ships no
// tests/golden vectors for this path and cannot be compiled here (Eigen absent), so this restatement is
// pinned only by the reference's two example-script properties (tests/test_oracle_known_answers.py), by
// independent numpy checks of the restated Eigen routines, and — for everything that is not Eigen arithmetic —
// by the reference's own translation units compiled against an Eigen stand-in and compared with this
// restatement bit for bit (oracle/eigen_standin/standin.h, tests/test_reference_structure_pin.py).
//
// CPU restatement of the MAD-ICP hot path, following the reference files function by function and
// keeping the reference's data structure and threading (heap-allocated pointer tree, std::async build,
// `omp parallel for` over keyframes with per-thread adders) because it doubles as the timed CPU baseline.
//
//   MADtree        <- mad_icp/src/tools/mad_tree.h:47-102, mad_tree.cpp:35-172, utils.h:37-97
//   MADicp         <- mad_icp/src/odometry/mad_icp.h:41-79, mad_icp.cpp:31-117
//   VelEstimator   <- mad_icp/src/odometry/vel_estimator.h:39-58, vel_estimator.cpp:32-97
//   Pipeline       <- mad_icp/src/odometry/pipeline.h:45-103, pipeline.cpp:34-308
//   Frame          <- mad_icp/src/tools/frame.h:37-52 ; constants <- tools/constants.h:31-35
//   MADtreeTool / MADicpTool <- pybind/tools/mad_tree_wrapper.h:34-71, mad_icp_wrapper.h:33-112
#pragma once
#include "linalg.h"

#include <cstdint>
#include <deque>
#include <memory>
#include <vector>

namespace oracle {

// tools/constants.h:31-35
static constexpr int CHUNKS = 1024;
static constexpr int SMOOTHING_T = 10;
static constexpr double E_THRESHOLD_VEL = 0.3162;
static constexpr int MAX_ICP_ITS = 15;
static constexpr int FRAME_WINDOW = 10;

using ContainerType = std::vector<Vec3>;  // mad_tree.h:42
using IteratorType = ContainerType::iterator;
struct MADtree;
using LeafList = std::vector<MADtree*>;  // mad_tree.h:45

struct MADtree {  // mad_tree.h:47-102
  MADtree(ContainerType* vec, IteratorType begin, IteratorType end, double b_max, double b_min, int level,
          int max_parallel_level, MADtree* parent, MADtree* plane_predecessor);
  ~MADtree() {
    delete left_;
    delete right_;
  }
  MADtree(const MADtree&) = delete;
  MADtree& operator=(const MADtree&) = delete;

  void build(ContainerType* vec, IteratorType begin, IteratorType end, double b_max, double b_min, int level,
             int max_parallel_level, MADtree* parent, MADtree* plane_predecessor);
  const MADtree* bestMatchingLeafFast(const Vec3& query) const;
  // same descent, also reports the number of internal nodes visited (for the algorithmic-bytes figure)
  const MADtree* bestMatchingLeafFastDepth(const Vec3& query, int& depth) const;
  void getLeafs(LeafList& out);
  void applyTransform(const Mat3& r, const Vec3& t);

  int num_points_ = 0;
  bool matched_ = false;  // reference leaves this uninitialised (mad_tree.h:92, quirk Q3); zero-init here
  MADtree* left_ = nullptr;
  MADtree* right_ = nullptr;
  MADtree* parent_ = nullptr;
  Vec3 mean_;
  Vec3 bbox_;
  Mat3 eigenvectors_;
};

struct Frame {  // frame.h:37-52
  Iso3 frame_to_map_ = Iso3::Identity();
  MADtree* tree_ = nullptr;
  LeafList leaves_;
  double stamp_ = 0.;
  double weight_ = 0.;
  int frame_ = 0;
};

class MADicp {  // mad_icp.h:41-79
 public:
  MADicp(double min_ball, double rho_ker, double b_ratio, int num_threads);
  void resetAdders();
  void setMoving(const LeafList& moving_leaves);
  void init(const Iso3& moving_in_fixed);
  void update(const MADtree* fixed_tree);
  void updateState();
  void errorAndJacobian(double& e, double J[6], const MADtree& fixed, const MADtree& moving,
                        const Vec3& moving_transformed) const;

  Iso3 X_;
  Mat6 H_adder_;
  Vec6 b_adder_;
  LeafList moving_leaves_;
  std::vector<Mat6> H_adders_;
  std::vector<Vec6> b_adders_;
  double rho_ker_;
  double min_ball_;
  double b_ratio_;
  int num_threads_;
  // instrumentation (not in the reference): sum of internal nodes visited by update(), per thread
  std::vector<long long> depth_adders_;
  // optional correspondence trace for parity tests (not in the reference): update() records, for moving
  // leaf i, the NN leaf it descended to and whether the gate at mad_icp.cpp:81-83 rejected it.
  struct Trace {
    std::vector<const MADtree*> nn;
    std::vector<uint8_t> rejected;
  };
  Trace* trace_ = nullptr;
};

struct VelEstimator {  // vel_estimator.h:39-58
  explicit VelEstimator(double sensor_hz);
  void setOdometry(const std::vector<Iso3>& odometry);
  void init(const Vec6& velocity);
  void oneRound();
  void errorAndJacobian(Vec6& e, const Iso3& T_now, const Iso3& T_prev, double delta_t);
  void update(const Iso3& T_now, const Iso3& T_prev, double delta_t, double weight);

  Mat6 H_adder_;
  Vec6 X_, b_adder_;
  std::vector<Iso3> odometry_;
  double ts_;
};

class Pipeline {  // pipeline.h:45-103
 public:
  Pipeline(double sensor_hz, bool deskew, double b_max, double rho_ker, double p_th, double b_min, double b_ratio,
           int num_keyframes, int num_threads, bool realtime);
  ~Pipeline();

  const Iso3& currentPose() const { return frame_to_map_; }
  const std::vector<Iso3>& trajectory() const { return trajectory_; }
  const Iso3& keyframePose() const { return keyframe_to_map_; }
  bool isInitialized() const { return is_initialized_; }
  size_t currentID() const { return seq_; }
  size_t keyframeID() const { return seq_keyframe_; }
  bool isMapUpdated() const { return is_map_updated_; }
  ContainerType currentLeaves() const;
  ContainerType modelLeaves() const;
  void compute(const double& curr_stamp, ContainerType curr_cloud_mem);

  // instrumentation (not in the reference)
  // test seam for realtime = true (tests/test_boundary.py): when virtual_pre_ms_ >= 0 the wall clock of pipeline.cpp:160-169,
  // 186-191 is replaced by a model — preprocessing took virtual_pre_ms_, every round virtual_round_ms_ — so that the
  // reference's per-round budget check becomes a deterministic function the product's round-count rule can be held to
  double virtual_pre_ms_ = -1.0, virtual_round_ms_ = 0.0;
  int last_rounds_ = 0;              // GN rounds the last frame ran
  double last_icp_ms_ = 0.0;         // wall time of the GN loop, the region the reference itself times
  double last_inliers_ratio_ = 0.0;  // pipeline.cpp:204
  size_t numKeyframes() const { return keyframes_.size(); }
  Iso3 last_guess_ = Iso3::Identity();  // the constant-velocity prediction the last frame's loop started from (pipeline.cpp:146-152)
  const MADtree* keyframeTree(size_t k) const { return k < keyframes_.size() ? keyframes_[k]->tree_ : nullptr; }
  Iso3 predict() const;  // the prediction the NEXT frame's loop would start from, without running it (pipeline.cpp:146-152)

 protected:
  void initialize(const double& curr_stamp, ContainerType* curr_cloud);
  void deskew(ContainerType* curr_cloud, const Iso3& T_prev, const Iso3& T_now);

  MADicp icp_;
  VelEstimator vel_estimator_;
  Iso3 frame_to_map_;
  Iso3 keyframe_to_map_;
  Vec6 current_velocity_;
  std::deque<Frame*> keyframes_;
  std::deque<Frame*> frames_;
  std::vector<Iso3> trajectory_;
  MADtree* current_tree_;
  LeafList model_leaves_, current_leaves_;
  bool deskew_, realtime_;
  int num_keyframes_, num_threads_, max_parallel_levels_;
  double sensor_hz_, b_max_, p_th_, b_min_;
  size_t seq_;
  size_t seq_keyframe_;
  bool is_initialized_;
  bool is_map_updated_;
  float loop_time;
};

// pybind/tools/mad_tree_wrapper.h:34-71
class MADtreeTool {
 public:
  void build(ContainerType vec, double b_max, double b_min, int max_parallel_level);
  const MADtree* search(const Vec3& query) const;
  const MADtree* root() const { return mad_tree_.get(); }

 protected:
  std::unique_ptr<MADtree> mad_tree_;
};

// pybind/tools/mad_icp_wrapper.h:33-112.  Quirks fixed as SURVEY §8 Q2/Q4 decide: the num_threads
// argument is used (the reference self-initialises the member), query leaves are cleared on re-set.
class MADicpTool {
 public:
  explicit MADicpTool(int num_threads);
  void setQueryCloud(ContainerType query, double b_max, double b_min);
  void setReferenceCloud(ContainerType reference, double b_max, double b_min);
  Iso3 compute(const Iso3& T, size_t max_icp_iterations, double rho_ker, double b_ratio);
  const MADtree* refTree() const { return ref_tree_.get(); }
  const MADtree* queryTree() const { return query_tree_.get(); }
  const LeafList& queryLeaves() const { return query_leaves_; }
  MADicp* icp() { return mad_icp_.get(); }

 protected:
  std::unique_ptr<MADicp> mad_icp_;
  std::unique_ptr<MADtree> ref_tree_;
  std::unique_ptr<MADtree> query_tree_;
  LeafList query_leaves_;
  double ref_b_max_ = 0.2;
  int max_parallel_levels_;
  int num_threads_;
};

}  // namespace oracle
