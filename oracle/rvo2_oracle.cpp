// ORACLE — TEST INFRASTRUCTURE ONLY (see rvo2_oracle.hpp header).
// float32 restatement of RVO2's agent-agent ORCA; build with -O2 -ffp-contract=off.
// Section tags (A.x) refer to SURVEY.md Appendix A.
#include "rvo2_oracle.hpp"

#include <algorithm>
#include <cmath>

namespace rvo2_oracle {

namespace {

constexpr float kEps = 0.00001f;  // RVO_EPSILON

inline Vec2 add(Vec2 a, Vec2 b) { return Vec2(a.x + b.x, a.y + b.y); }
inline Vec2 sub(Vec2 a, Vec2 b) { return Vec2(a.x - b.x, a.y - b.y); }
inline Vec2 neg(Vec2 a) { return Vec2(-a.x, -a.y); }
inline Vec2 scale(float s, Vec2 a) { return Vec2(s * a.x, s * a.y); }
inline float dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
inline float det(Vec2 a, Vec2 b) { return a.x * b.y - a.y * b.x; }
inline float absSq(Vec2 a) { return dot(a, a); }
inline float length(Vec2 a) { return std::sqrt(dot(a, a)); }
// A.0: vector / scalar is a multiply by the reciprocal.
inline Vec2 divide(Vec2 a, float s) {
    const float inv = 1.0f / s;
    return Vec2(a.x * inv, a.y * inv);
}
inline Vec2 normalize(Vec2 a) { return divide(a, length(a)); }
inline float sqr(float v) { return v * v; }

}  // namespace

// ---------------------------------------------------------------- linear programs (A.4 - A.6)

bool lp1(const std::vector<HalfPlane>& lines, std::size_t lineNo, float radius, Vec2 opt,
         bool directionOpt, Vec2& result) {
    const HalfPlane& L = lines[lineNo];
    const float dp = dot(L.point, L.direction);
    const float disc = sqr(dp) + sqr(radius) - absSq(L.point);
    if (disc < 0.0f) return false;  // speed disc misses the line entirely

    const float root = std::sqrt(disc);
    float tLeft = -dp - root;
    float tRight = -dp + root;

    for (std::size_t i = 0; i < lineNo; ++i) {
        const float den = det(L.direction, lines[i].direction);
        const float num = det(lines[i].direction, sub(L.point, lines[i].point));
        if (std::fabs(den) <= kEps) {  // (almost) parallel
            if (num < 0.0f) return false;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) {
            tRight = std::min(tRight, t);
        } else {
            tLeft = std::max(tLeft, t);
        }
        if (tLeft > tRight) return false;
    }

    if (directionOpt) {
        if (dot(opt, L.direction) > 0.0f) {
            result = add(L.point, scale(tRight, L.direction));
        } else {
            result = add(L.point, scale(tLeft, L.direction));
        }
    } else {
        const float t = dot(L.direction, sub(opt, L.point));
        if (t < tLeft) {
            result = add(L.point, scale(tLeft, L.direction));
        } else if (t > tRight) {
            result = add(L.point, scale(tRight, L.direction));
        } else {
            result = add(L.point, scale(t, L.direction));
        }
    }
    return true;
}

std::size_t lp2(const std::vector<HalfPlane>& lines, float radius, Vec2 opt, bool directionOpt,
                Vec2& result) {
    if (directionOpt) {
        result = Vec2(opt.x * radius, opt.y * radius);
    } else if (absSq(opt) > sqr(radius)) {
        const Vec2 n = normalize(opt);
        result = Vec2(n.x * radius, n.y * radius);
    } else {
        result = opt;
    }
    for (std::size_t i = 0; i < lines.size(); ++i) {
        if (det(lines[i].direction, sub(lines[i].point, result)) > 0.0f) {
            const Vec2 keep = result;
            if (!lp1(lines, i, radius, opt, directionOpt, result)) {
                result = keep;
                return i;
            }
        }
    }
    return lines.size();
}

void lp3(const std::vector<HalfPlane>& lines, std::size_t numObstLines, std::size_t beginLine,
         float radius, Vec2& result) {
    float distance = 0.0f;
    for (std::size_t i = beginLine; i < lines.size(); ++i) {
        if (det(lines[i].direction, sub(lines[i].point, result)) > distance) {
            std::vector<HalfPlane> proj(lines.begin(),
                                        lines.begin() + static_cast<std::ptrdiff_t>(numObstLines));
            for (std::size_t j = numObstLines; j < i; ++j) {
                HalfPlane h;
                const float d = det(lines[i].direction, lines[j].direction);
                if (std::fabs(d) <= kEps) {
                    if (dot(lines[i].direction, lines[j].direction) > 0.0f) continue;  // same way
                    h.point = scale(0.5f, add(lines[i].point, lines[j].point));
                } else {
                    const float t =
                        det(lines[j].direction, sub(lines[i].point, lines[j].point)) / d;
                    h.point = add(lines[i].point, scale(t, lines[i].direction));
                }
                h.direction = normalize(sub(lines[j].direction, lines[i].direction));
                proj.push_back(h);
            }
            const Vec2 keep = result;
            const Vec2 optDir(-lines[i].direction.y, lines[i].direction.x);
            if (lp2(proj, radius, optDir, true, result) < proj.size()) {
                result = keep;  // numerical corner: keep the previous answer
            }
            distance = det(lines[i].direction, sub(lines[i].point, result));
        }
    }
}

// ---------------------------------------------------------------- simulator

Simulator::Simulator(float timeStep, float neighborDist, std::size_t maxNeighbors,
                     float timeHorizon, float timeHorizonObst, float radius, float maxSpeed)
    : timeStep_(timeStep),
      defNeighborDist_(neighborDist),
      defTimeHorizon_(timeHorizon),
      defTimeHorizonObst_(timeHorizonObst),
      defRadius_(radius),
      defMaxSpeed_(maxSpeed),
      defMaxNeighbors_(maxNeighbors) {}

std::size_t Simulator::addAgent(float px, float py, float neighborDist, std::size_t maxNeighbors,
                                float timeHorizon, float timeHorizonObst, float radius,
                                float maxSpeed, float vx, float vy) {
    AgentRec a;
    a.position = Vec2(px, py);
    a.velocity = Vec2(vx, vy);
    a.neighborDist = neighborDist;
    a.maxNeighbors = maxNeighbors;
    a.timeHorizon = timeHorizon;
    a.timeHorizonObst = timeHorizonObst;
    a.radius = radius;
    a.maxSpeed = maxSpeed;
    agents_.push_back(a);
    return agents_.size() - 1;
}

// A.2 — kd-tree over a persistent permutation; leaves of at most 10 agents.
void Simulator::buildTree() {
    if (order_.size() < agents_.size()) {
        for (std::size_t i = order_.size(); i < agents_.size(); ++i) order_.push_back(i);
        tree_.assign(2 * order_.size() - 1, TreeNode());
    }
    if (!order_.empty()) buildTreeRec(0, order_.size(), 0);
}

void Simulator::buildTreeRec(std::size_t begin, std::size_t end, std::size_t node) {
    TreeNode& n = tree_[node];
    n.begin = begin;
    n.end = end;
    n.minX = n.maxX = agents_[order_[begin]].position.x;
    n.minY = n.maxY = agents_[order_[begin]].position.y;
    for (std::size_t i = begin + 1; i < end; ++i) {
        const Vec2 p = agents_[order_[i]].position;
        n.maxX = std::max(n.maxX, p.x);
        n.minX = std::min(n.minX, p.x);
        n.maxY = std::max(n.maxY, p.y);
        n.minY = std::min(n.minY, p.y);
    }
    if (end - begin > kMaxLeaf) {
        const bool vertical = (n.maxX - n.minX > n.maxY - n.minY);
        const float split = vertical ? 0.5f * (n.maxX + n.minX) : 0.5f * (n.maxY + n.minY);
        std::size_t left = begin, right = end;
        auto coord = [&](std::size_t k) {
            const Vec2 p = agents_[order_[k]].position;
            return vertical ? p.x : p.y;
        };
        while (left < right) {
            while (left < right && coord(left) < split) ++left;
            while (right > left && coord(right - 1) >= split) --right;
            if (left < right) {
                std::swap(order_[left], order_[right - 1]);
                ++left;
                --right;
            }
        }
        if (left == begin) {
            ++left;
            ++right;
        }
        const std::size_t l = node + 1;
        const std::size_t r = node + 2 * (left - begin);
        tree_[node].left = l;
        tree_[node].right = r;
        buildTreeRec(begin, left, l);
        buildTreeRec(left, end, r);
    }
}

void Simulator::offerNeighbor(AgentRec& a, std::size_t selfId, std::size_t otherId,
                              float& rangeSq) const {
    if (selfId == otherId) return;
    const float distSq = absSq(sub(a.position, agents_[otherId].position));
    if (distSq < rangeSq) {
        if (a.neighbors.size() < a.maxNeighbors) a.neighbors.emplace_back(distSq, otherId);
        std::size_t i = a.neighbors.size() - 1;
        while (i != 0 && distSq < a.neighbors[i - 1].first) {
            a.neighbors[i] = a.neighbors[i - 1];
            --i;
        }
        a.neighbors[i] = std::make_pair(distSq, otherId);
        if (a.neighbors.size() == a.maxNeighbors) rangeSq = a.neighbors.back().first;
    }
}

void Simulator::queryTreeRec(AgentRec& a, std::size_t selfId, float& rangeSq,
                             std::size_t node) const {
    const TreeNode& n = tree_[node];
    if (n.end - n.begin <= kMaxLeaf) {
        for (std::size_t i = n.begin; i < n.end; ++i) offerNeighbor(a, selfId, order_[i], rangeSq);
        return;
    }
    auto boxDistSq = [&](const TreeNode& c) {
        return sqr(std::max(0.0f, c.minX - a.position.x)) +
               sqr(std::max(0.0f, a.position.x - c.maxX)) +
               sqr(std::max(0.0f, c.minY - a.position.y)) +
               sqr(std::max(0.0f, a.position.y - c.maxY));
    };
    const float dL = boxDistSq(tree_[n.left]);
    const float dR = boxDistSq(tree_[n.right]);
    if (dL < dR) {
        if (dL < rangeSq) {
            queryTreeRec(a, selfId, rangeSq, n.left);
            if (dR < rangeSq) queryTreeRec(a, selfId, rangeSq, n.right);
        }
    } else {
        if (dR < rangeSq) {
            queryTreeRec(a, selfId, rangeSq, n.right);
            if (dL < rangeSq) queryTreeRec(a, selfId, rangeSq, n.left);
        }
    }
}

void Simulator::collectNeighbors(std::size_t i) {
    AgentRec& a = agents_[i];
    a.neighbors.clear();
    if (a.maxNeighbors > 0) {
        float rangeSq = sqr(a.neighborDist);
        queryTreeRec(a, i, rangeSq, 0);
    }
}

// A.3 + A.7 — half-plane per neighbour, then LP2 (+ LP3 on infeasibility).
void Simulator::computeNewVelocity(std::size_t i) {
    AgentRec& a = agents_[i];
    a.lines.clear();
    const std::size_t numObstLines = 0;  // the reference never adds obstacles
    const float invTimeHorizon = 1.0f / a.timeHorizon;

    for (const auto& nb : a.neighbors) {
        const AgentRec& o = agents_[nb.second];
        const Vec2 relPos = sub(o.position, a.position);
        const Vec2 relVel = sub(a.velocity, o.velocity);
        const float distSq = absSq(relPos);
        const float R = a.radius + o.radius;
        const float RSq = sqr(R);

        HalfPlane line;
        Vec2 u;
        if (distSq > RSq) {
            const Vec2 w = sub(relVel, scale(invTimeHorizon, relPos));
            const float wLenSq = absSq(w);
            const float dot1 = dot(w, relPos);
            if (dot1 < 0.0f && sqr(dot1) > RSq * wLenSq) {
                // closest point is on the cut-off disc
                const float wLen = std::sqrt(wLenSq);
                const Vec2 unitW = divide(w, wLen);
                line.direction = Vec2(unitW.y, -unitW.x);
                u = scale(R * invTimeHorizon - wLen, unitW);
            } else {
                // closest point is on one of the legs
                const float leg = std::sqrt(distSq - RSq);
                if (det(relPos, w) > 0.0f) {
                    line.direction = divide(
                        Vec2(relPos.x * leg - relPos.y * R, relPos.x * R + relPos.y * leg), distSq);
                } else {
                    line.direction = neg(divide(
                        Vec2(relPos.x * leg + relPos.y * R, -relPos.x * R + relPos.y * leg),
                        distSq));
                }
                const float dot2 = dot(relVel, line.direction);
                u = sub(scale(dot2, line.direction), relVel);
            }
        } else {
            // already overlapping: resolve within one time step
            const float invTimeStep = 1.0f / timeStep_;
            const Vec2 w = sub(relVel, scale(invTimeStep, relPos));
            const float wLen = length(w);
            const Vec2 unitW = divide(w, wLen);
            line.direction = Vec2(unitW.y, -unitW.x);
            u = scale(R * invTimeStep - wLen, unitW);
        }
        line.point = add(a.velocity, scale(0.5f, u));
        a.lines.push_back(line);
    }

    const std::size_t fail = lp2(a.lines, a.maxSpeed, a.prefVelocity, false, a.newVelocity);
    if (fail < a.lines.size()) lp3(a.lines, numObstLines, fail, a.maxSpeed, a.newVelocity);
}

void Simulator::solveOnly(std::size_t i) {
    buildTree();
    collectNeighbors(i);
    computeNewVelocity(i);
}

void Simulator::doStep() {
    buildTree();
    for (std::size_t i = 0; i < agents_.size(); ++i) {
        collectNeighbors(i);
        computeNewVelocity(i);
    }
    for (AgentRec& a : agents_) {
        a.velocity = a.newVelocity;
        a.position = add(a.position, scale(timeStep_, a.velocity));  // position += velocity * dt
    }
    globalTime_ += timeStep_;
}

}  // namespace rvo2_oracle
