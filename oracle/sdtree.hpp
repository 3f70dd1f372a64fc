// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header).
// CPU restatement of pt/src/integrator/path_guiding/dtree.rs (whole file) and stree.rs (whole file),
// plus square_to_dir / dir_to_square from path_guiding/mod.rs:404-421.
// The reference guards each collector with a Mutex; the oracle is run single-threaded per wave pass or
// with a lock per leaf (see integrators.hpp), so plain structs suffice here.
#pragma once
#include <cstring>
#include <mutex>
#include <vector>

#include "sampling.hpp"

namespace orc {

enum class Filter : int32_t { Nearest = 0, Box = 1, Stochastic = 2 };

inline V3 square_to_dir(V2 p) {  // mod.rs:404-410
    float cos_theta = 2.0f * p.x - 1.0f;
    float phi = TWO_PI * p.y;
    float sin_theta = safe_sqrt(1.0f - cos_theta * cos_theta);
    float sin_phi = std::sin(phi), cos_phi = std::cos(phi);
    return v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
}
inline V2 dir_to_square(V3 dir) {  // mod.rs:412-421
    float cos_theta = rclamp(dir.z, -1.0f, 1.0f);
    float phi = std::atan2(dir.y, dir.x);
    while (phi < 0.0f) phi += TWO_PI;
    V2 p{(cos_theta + 1.0f) / 2.0f, phi * INV_TWO_PI};
    return v2clamp(p, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
}

struct DTreeNode {  // dtree.rs:28-32,354-376
    float radiance[4] = {0, 0, 0, 0};
    uint16_t children[4] = {0, 0, 0, 0};
    float sum() const {
        float s = 0.0f;
        for (int i = 0; i < 4; i++) s += radiance[i];
        return s;
    }
    bool is_leaf(int c) const { return children[c] == 0; }
};

inline int dtree_child_index(V2 p, B2 bounds, B2* child_bounds) {  // dtree.rs:331-352
    V2 mid = bounds.centroid();
    V2 mn = bounds.min, mx = bounds.max;
    int res = 0;
    if (p.x < mid.x) mx.x = mid.x;
    else {
        mn.x = mid.x;
        res |= 1;
    }
    if (p.y < mid.y) mx.y = mid.y;
    else {
        mn.y = mid.y;
        res |= 2;
    }
    *child_bounds = B2{mn, mx};
    return res;
}
inline B2 dtree_child_bounds(B2 b, int child) {  // dtree.rs:119-139 / 245-265
    V2 mn = b.min, md = b.centroid(), mx = b.max;
    switch (child) {
        case 0: return B2{{mn.x, mn.y}, {md.x, md.y}};
        case 1: return B2{{md.x, mn.y}, {mx.x, md.y}};
        case 2: return B2{{mn.x, md.y}, {md.x, mx.y}};
        default: return B2{{md.x, md.y}, {mx.x, mx.y}};
    }
}

struct DTree {
    float statistical_weight = 0.0f;
    Filter filter = Filter::Nearest;
    std::vector<DTreeNode> nodes;
    std::vector<uint16_t> free_slots;

    static DTree new_root(Filter f) {
        DTree d;
        d.filter = f;
        d.nodes.push_back(DTreeNode{});
        return d;
    }
    static DTree no_alloc() { return DTree{}; }
    bool is_empty() const { return nodes.empty(); }
    float sum() const { return nodes[0].sum(); }
    float mean() const { return statistical_weight == 0.0f ? 0.0f : sum() / (FOUR_PI * statistical_weight); }
    B2 leaf_bounds(V2 p) const {  // dtree.rs:81-94
        B2 bounds = B2::unit();
        uint16_t ni = 0;
        for (;;) {
            const DTreeNode& node = nodes[ni];
            B2 cb;
            int child = dtree_child_index(p, bounds, &cb);
            if (node.is_leaf(child)) return cb;
            ni = node.children[child];
            bounds = cb;
        }
    }
    // ---- sampling side (dtree.rs:97-180)
    V2 sample(Sampler& sampler) const {
        if (mean() == 0.0f) {
            float u[2];
            sampler.sample2d(u);
            return V2{u[0], u[1]};
        }
        uint16_t ni = 0;
        B2 bounds = B2::unit();
        V2 s;
        for (;;) {
            const DTreeNode& node = nodes[ni];
            int child = (int)sample_discrete(sampler.sample1d(), node.radiance, 4);
            B2 cb = dtree_child_bounds(bounds, child);
            if (node.is_leaf(child)) {
                float u[2];
                sampler.sample2d(u);
                V2 smp = cb.lerp(V2{u[0], u[1]});
                V2 lim = cb.max - V2{F32_EPSILON, F32_EPSILON};
                s = v2min(smp, lim);
                break;
            }
            ni = node.children[child];
            bounds = cb;
        }
        return v2clamp(s, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
    }
    float pdf(V2 p) const {
        if (mean() == 0.0f) return INV_FOUR_PI;
        // pdf_recursive multiplies scale_outer * (scale_inner * (...)) — evaluate innermost first.
        float scales[64];
        int n = 0;
        uint16_t ni = 0;
        B2 bounds = B2::unit();
        float result;
        for (;;) {
            const DTreeNode& node = nodes[ni];
            B2 cb;
            int child = dtree_child_index(p, bounds, &cb);
            if (node.radiance[child] == 0.0f) {
                result = 0.0f;
                break;
            }
            float scale = 4.0f * node.radiance[child] / node.sum();
            if (node.is_leaf(child)) {
                result = scale;
                break;
            }
            scales[n++] = scale;
            ni = node.children[child];
            bounds = cb;
        }
        for (int i = n - 1; i >= 0; i--) result = scales[i] * result;
        return result * INV_FOUR_PI;
    }
    // ---- collecting side (dtree.rs:182-329)
    void build_recursive(uint16_t ni) {
        for (int child = 0; child < 4; child++) {
            if (nodes[ni].is_leaf(child)) continue;
            uint16_t cn = nodes[ni].children[child];
            build_recursive(cn);
            nodes[ni].radiance[child] = nodes[cn].sum();
        }
    }
    void build() { build_recursive(0); }
    void record(V2 p, float radiance, float w, Sampler& sampler) {
        statistical_weight += w;
        switch (filter) {
            case Filter::Nearest: record_nearest(p, radiance); break;
            case Filter::Box: record_box(p, radiance); break;
            case Filter::Stochastic: record_stochastic(p, radiance, sampler); break;
        }
    }
    void record_nearest(V2 p, float radiance) {
        B2 bounds = B2::unit();
        uint16_t ni = 0;
        for (;;) {
            DTreeNode& node = nodes[ni];
            B2 cb;
            int child = dtree_child_index(p, bounds, &cb);
            if (node.is_leaf(child)) {
                node.radiance[child] += radiance;
                return;
            }
            ni = node.children[child];
            bounds = cb;
        }
    }
    void record_stochastic(V2 p, float radiance, Sampler& sampler) {
        B2 nb = leaf_bounds(p);
        B2 jitter = B2::around(p, nb.extent());
        float u0 = sampler.sample1d(), u1 = sampler.sample1d();
        V2 q = B2::unit().clip(jitter.lerp(V2{u0, u1}));
        record_nearest(q, radiance);
    }
    void record_box_recursive(B2 node_bounds, B2 footprint, float radiance, uint16_t ni) {
        for (int child = 0; child < 4; child++) {
            B2 cb = dtree_child_bounds(node_bounds, child);
            float w = footprint.overlap(cb) / footprint.area();
            if (w == 0.0f) continue;
            if (nodes[ni].is_leaf(child)) nodes[ni].radiance[child] += radiance * w;
            else record_box_recursive(cb, footprint, radiance, nodes[ni].children[child]);
        }
    }
    void record_box(V2 p, float radiance) {
        B2 lb = leaf_bounds(p);
        B2 footprint = B2::around(p, lb.extent()).clamp(V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
        record_box_recursive(B2::unit(), footprint, radiance, 0);
    }
    // NodeList (dtree.rs:378-418)
    uint16_t add_node(const DTreeNode& n) {
        if (!free_slots.empty()) {
            uint16_t idx = free_slots.back();
            free_slots.pop_back();
            nodes[idx] = n;
            return idx;
        }
        uint16_t idx = (uint16_t)nodes.size();
        nodes.push_back(n);
        return idx;
    }
    void free_recursive(uint16_t ni) {
        if (ni == 0) return;
        free_slots.push_back(ni);
        DTreeNode node = nodes[ni];
        for (int c = 0; c < 4; c++)
            if (!node.is_leaf(c)) free_recursive(node.children[c]);
    }
    void subdivide_and_clear(float threshold) {  // dtree.rs:291-324
        std::vector<std::pair<uint16_t, int>> stack;
        stack.push_back({0, 1});
        float total = sum();
        while (!stack.empty()) {
            auto [ni, depth] = stack.back();
            stack.pop_back();
            DTreeNode node = nodes[ni];  // COPY, as in the reference
            for (int child = 0; child < 4; child++) {
                float frac;
                if (total > 0.0f) frac = node.radiance[child] / total;
                else {
                    frac = 1.0f;  // 0.25f32.powi(depth): exact power of two
                    for (int i = 0; i < depth; i++) frac *= 0.25f;
                }
                if (frac < threshold) {
                    free_recursive(node.children[child]);
                    nodes[ni].children[child] = 0;
                    continue;
                }
                if (node.is_leaf(child)) {
                    DTreeNode nb;
                    for (int i = 0; i < 4; i++) nb.radiance[i] = node.radiance[child] / 4.0f;
                    uint16_t idx = add_node(nb);
                    nodes[ni].children[child] = idx;
                    stack.push_back({idx, depth + 1});
                } else {
                    stack.push_back({node.children[child], depth + 1});
                }
            }
        }
        statistical_weight = 0.0f;
        for (auto& n : nodes)
            for (int i = 0; i < 4; i++) n.radiance[i] = 0.0f;
    }
    void emit_into(DTree& sampler) const {  // dtree.rs:325-328 (filter of `sampler` untouched)
        sampler.statistical_weight = statistical_weight;
        sampler.nodes = nodes;
        sampler.free_slots = free_slots;
    }
};

// Test hook (tests/test_gpu_sdtree.py): the arguments of every Guider::add / STree::record call that passes the early
// return (mod.rs:106-115, stree.rs:102-116) with the stochastic filter's three draws and the leaf that received it,
// keyed by the path that made it. 14 words per record: p, wi, radiance, u0, u1, u2, leaf NODE, pixel, pass, vertex.
struct RecordLog {
    std::mutex m;
    std::vector<uint32_t> words;
};
struct RecordKey {
    uint32_t pixel = 0, pass = 0, vertex = 0;
};
inline RecordKey& record_key() {  // set by guided_wave_passes before each record() call of its thread
    thread_local RecordKey k;
    return k;
}

struct STreeNode {  // stree.rs:12-19
    DTree sampler, collector;
    size_t left = 0, right = 0, split_axis = 0, depth = 0;
    std::unique_ptr<std::mutex> lock = std::make_unique<std::mutex>();
    bool is_leaf() const { return left == 0; }
    size_t child_node(size_t i) const { return i == 0 ? left : right; }
};

struct STree {
    std::vector<STreeNode> nodes;
    B3 bounds;
    Filter filter = Filter::Stochastic;
    RecordLog* log = nullptr;  // not owned

    void init(B3 scene_bounds, Filter spatial, Filter directional) {  // stree.rs:21-34
        nodes.clear();
        STreeNode root;
        root.sampler = DTree::new_root(directional);
        root.collector = DTree::new_root(directional);
        nodes.push_back(std::move(root));
        bounds = B3::around(scene_bounds.centroid(), splat3(max_component(scene_bounds.extent())));
        filter = spatial;
    }
    size_t leaf(V3 p, B3* leaf_bounds) const {  // stree.rs:202-229
        B3 b = bounds;
        size_t ni = 0;
        for (;;) {
            const STreeNode& node = nodes[ni];
            if (node.is_leaf()) {
                *leaf_bounds = b;
                return ni;
            }
            V3 mid = b.centroid();
            size_t axis = node.split_axis;
            size_t child;
            if (p[(int)axis] < mid[(int)axis]) {
                b.max.at((int)axis) = mid[(int)axis];
                child = 0;
            } else {
                b.min.at((int)axis) = mid[(int)axis];
                child = 1;
            }
            ni = node.child_node(child);
        }
    }
    const DTree& sampler(V3 p) const {
        B3 lb;
        return nodes[leaf(p, &lb)].sampler;
    }
    void node_record(size_t ni, V2 p2, float radiance, float w, Sampler& sampler) {
        std::lock_guard<std::mutex> g(*nodes[ni].lock);
        nodes[ni].collector.record(p2, radiance, w, sampler);
    }
    size_t record_nearest(V3 p3, V2 p2, float radiance, Sampler& sampler) {
        B3 lb;
        size_t ni = leaf(p3, &lb);
        node_record(ni, p2, radiance, 1.0f, sampler);
        return ni;
    }
    void record_box_recursive(B3 node_bounds, B3 footprint, V2 p2, float radiance, size_t ni, Sampler& sampler) {  // stree.rs:134-172
        for (size_t i = 0; i < 2; i++) {
            const STreeNode& node = nodes[ni];
            int ax = (int)node.split_axis;
            float mid = node_bounds.centroid()[ax];
            V3 mn = node_bounds.min, mx = node_bounds.max;
            if (i == 0) mx.at(ax) = mid;
            else mn.at(ax) = mid;
            B3 cb{mn, mx};
            float w = footprint.overlap(cb) / footprint.volume();
            if (w == 0.0f) continue;
            if (node.is_leaf()) node_record(ni, p2, radiance * w, w, sampler);
            else record_box_recursive(cb, footprint, p2, radiance, node.child_node(i), sampler);
        }
    }
    void record_box(V3 p3, V2 p2, float radiance, Sampler& sampler) {
        B3 lb;
        leaf(p3, &lb);
        B3 footprint = B3::around(p3, lb.extent()).clamp(bounds.min, bounds.max);
        record_box_recursive(bounds, footprint, p2, radiance, 0, sampler);
    }
    void log_record(V3 p3, V3 wi, float radiance, float u0, float u1, float u2, size_t ni) {  // test hook (RecordLog)
        if (!log) return;
        const RecordKey k = record_key();
        const float f[10] = {p3.x, p3.y, p3.z, wi.x, wi.y, wi.z, radiance, u0, u1, u2};
        uint32_t w[14];
        std::memcpy(w, f, sizeof(f));
        w[10] = (uint32_t)ni; w[11] = k.pixel; w[12] = k.pass; w[13] = k.vertex;
        std::lock_guard<std::mutex> g(log->m);
        log->words.insert(log->words.end(), w, w + 14);
    }
    void record_stochastic(V3 p3, V3 wi, V2 p2, float radiance, Sampler& sampler) {  // stree.rs:186-201
        float u0 = sampler.sample1d(), u1 = sampler.sample1d(), u2 = sampler.sample1d();
        size_t ni = record_stochastic_drawn(p3, p2, radiance, u0, u1, u2, sampler);
        log_record(p3, wi, radiance, u0, u1, u2, ni);
    }
    // the body of record_stochastic after its three draws (split off so that a test can replay given draws); -> leaf node
    size_t record_stochastic_drawn(V3 p3, V2 p2, float radiance, float u0, float u1, float u2, Sampler& sampler) {
        B3 lb;
        leaf(p3, &lb);
        B3 jitter = B3::around(p3, lb.extent());
        V3 q = bounds.clip(jitter.lerp(v3(u0, u1, u2)));
        return record_nearest(q, p2, radiance, sampler);
    }
    // STree::record (stree.rs:102-116) with the stochastic filter's draws supplied by the caller (the other filters draw nothing)
    void record_drawn(V3 p, V3 wi, float radiance, float u0, float u1, float u2, Sampler& sampler) {
        if (!std::isfinite(radiance)) return;
        if (radiance == 0.0f) return;
        V2 p2 = dir_to_square(wi);
        switch (filter) {
            case Filter::Nearest: record_nearest(p, p2, radiance, sampler); break;
            case Filter::Box: record_box(p, p2, radiance, sampler); break;
            case Filter::Stochastic: record_stochastic_drawn(p, p2, radiance, u0, u1, u2, sampler); break;
        }
    }
    void record(V3 p, V3 wi, float radiance, Sampler& sampler) {  // stree.rs:102-116
        if (!std::isfinite(radiance)) return;
        if (radiance == 0.0f) return;
        V2 p2 = dir_to_square(wi);
        switch (filter) {
            case Filter::Nearest: log_record(p, wi, radiance, 0.0f, 0.0f, 0.0f, record_nearest(p, p2, radiance, sampler)); break;
            case Filter::Box: record_box(p, p2, radiance, sampler); break;
            case Filter::Stochastic: record_stochastic(p, wi, p2, radiance, sampler); break;
        }
    }
    void subdivide_nodes(float base, size_t iteration) {  // stree.rs:55-101
        float p2 = 1.0f;  // 2f32.powi(iteration)
        for (size_t k = 0; k < iteration; k++) p2 *= 2.0f;
        float threshold = base * std::sqrt(p2);
        size_t i = 0;
        while (i < nodes.size()) {
            size_t lidx = nodes.size(), ridx = nodes.size() + 1;
            if (nodes[i].is_leaf() && nodes[i].collector.statistical_weight > threshold) {
                STreeNode left, right;
                left.sampler = nodes[i].sampler;
                left.collector = nodes[i].collector;
                left.split_axis = (nodes[i].split_axis + 1) % 3;
                left.depth = nodes[i].depth + 1;
                right.sampler = std::move(nodes[i].sampler);
                right.collector = std::move(nodes[i].collector);
                nodes[i].sampler = DTree::no_alloc();
                nodes[i].collector = DTree::no_alloc();
                right.split_axis = (nodes[i].split_axis + 1) % 3;
                right.depth = nodes[i].depth + 1;
                left.collector.statistical_weight /= 2.0f;
                right.collector.statistical_weight /= 2.0f;
                nodes[i].left = lidx;
                nodes[i].right = ridx;
                nodes.push_back(std::move(left));
                nodes.push_back(std::move(right));
            }
            i++;
        }
    }
    void new_iteration(float directional_threshold, float spatial_threshold, size_t iteration) {  // stree.rs:42-54
        subdivide_nodes(spatial_threshold, iteration);
        for (auto& node : nodes) {
            if (node.is_leaf()) {  // stree.rs:255-260
                node.collector.build();
                node.collector.emit_into(node.sampler);
                node.collector.subdivide_and_clear(directional_threshold);
            }
        }
    }
};

}  // namespace orc
