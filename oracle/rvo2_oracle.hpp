// ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
// (crowdnav_amd/, the C-ABI library).  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may use anything under oracle/.
//
// CPU float32 restatement of the agent-agent ORCA solver of the RVO2 library (v2.0.x), the
// un-vendored, un-pinned native dependency (`rvo2` = sybrenstuvel/Python-RVO2, mentioned only at
// reference README.md:29) that the reference calls at
//   crowd_sim/envs/policy/orca.py:99-129   (ctor, addAgent, setAgent*, doStep, getAgentVelocity)
//   crowd_sim/envs/crowd_sim.py:221-245    (get_human_times; out of scope)
// The source of that library is NOT on disk; this file follows the published algorithm
// (van den Berg et al., "Reciprocal n-body collision avoidance") with the arithmetic conventions
// listed in SURVEY.md Appendix A:  IEEE binary32 everywhere, no FMA contraction (build with
// -ffp-contract=off), vector/scalar division = multiply by the reciprocal, strict comparisons,
// RVO_EPSILON = 1e-5f, neighbour list kept sorted by insertion with strict '<'.
//
// PARITY STATUS: "parity unpinned" against upstream Python-RVO2 binaries (none available offline);
// pinned only by the aggregate anchor of SURVEY.md Appendix D (213/284/3 outcomes, 15 190 steps over
// the 500 test cases) and by the fixtures generated with the *unmodified reference Python* driving
// this module (oracle/gen_golden.py).
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace rvo2_oracle {

struct Vec2 {
    float x = 0.0f, y = 0.0f;
    Vec2() = default;
    Vec2(float x_, float y_) : x(x_), y(y_) {}
};

struct HalfPlane {  // feasible side = left of (point, direction)
    Vec2 point;
    Vec2 direction;
};

struct AgentRec {
    Vec2 position, velocity, prefVelocity, newVelocity;
    float neighborDist = 0, timeHorizon = 0, timeHorizonObst = 0, radius = 0, maxSpeed = 0;
    std::size_t maxNeighbors = 0;
    std::vector<std::pair<float, std::size_t>> neighbors;  // (distSq, agent id), ascending
    std::vector<HalfPlane> lines;
};

class Simulator {
public:
    Simulator(float timeStep, float neighborDist, std::size_t maxNeighbors, float timeHorizon,
              float timeHorizonObst, float radius, float maxSpeed);

    std::size_t addAgent(float px, float py, float neighborDist, std::size_t maxNeighbors,
                         float timeHorizon, float timeHorizonObst, float radius, float maxSpeed,
                         float vx, float vy);
    std::size_t numAgents() const { return agents_.size(); }
    void setPosition(std::size_t i, float x, float y) { agents_[i].position = Vec2(x, y); }
    void setVelocity(std::size_t i, float x, float y) { agents_[i].velocity = Vec2(x, y); }
    void setPrefVelocity(std::size_t i, float x, float y) { agents_[i].prefVelocity = Vec2(x, y); }
    Vec2 position(std::size_t i) const { return agents_[i].position; }
    Vec2 velocity(std::size_t i) const { return agents_[i].velocity; }
    float timeStep() const { return timeStep_; }
    float globalTime() const { return globalTime_; }

    // One simulator step: rebuild the kd-tree, solve every agent from the same snapshot, then
    // commit velocities and integrate positions.
    void doStep();
    // Solve a single agent only (used by the batched oracle: orca.py:129 consumes agent 0 only,
    // and the solve of agent i is a pure function of the pre-step snapshot).
    void solveOnly(std::size_t i);
    Vec2 newVelocity(std::size_t i) const { return agents_[i].newVelocity; }
    const AgentRec& agent(std::size_t i) const { return agents_[i]; }

private:
    struct TreeNode {
        std::size_t begin = 0, end = 0, left = 0, right = 0;
        float minX = 0, maxX = 0, minY = 0, maxY = 0;
    };
    static constexpr std::size_t kMaxLeaf = 10;

    void buildTree();
    void buildTreeRec(std::size_t begin, std::size_t end, std::size_t node);
    void queryTreeRec(AgentRec& a, std::size_t selfId, float& rangeSq, std::size_t node) const;
    void offerNeighbor(AgentRec& a, std::size_t selfId, std::size_t otherId, float& rangeSq) const;
    void collectNeighbors(std::size_t i);
    void computeNewVelocity(std::size_t i);

    float timeStep_;
    float globalTime_ = 0.0f;
    // defaults (kept for API completeness; the reference always passes explicit values)
    float defNeighborDist_, defTimeHorizon_, defTimeHorizonObst_, defRadius_, defMaxSpeed_;
    std::size_t defMaxNeighbors_;
    std::vector<AgentRec> agents_;
    std::vector<std::size_t> order_;  // persistent permutation the kd-tree partitions in place
    std::vector<TreeNode> tree_;
};

// The three incremental linear programs, exposed for unit tests.
bool lp1(const std::vector<HalfPlane>& lines, std::size_t lineNo, float radius, Vec2 opt,
         bool directionOpt, Vec2& result);
std::size_t lp2(const std::vector<HalfPlane>& lines, float radius, Vec2 opt, bool directionOpt,
                Vec2& result);
void lp3(const std::vector<HalfPlane>& lines, std::size_t numObstLines, std::size_t beginLine,
         float radius, Vec2& result);

}  // namespace rvo2_oracle
