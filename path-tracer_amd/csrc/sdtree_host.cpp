// sdtree_host.cpp — see sdtree_host.h. Host code only.
#include "sdtree_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <utility>

namespace pt {
namespace {

// The per-leaf work between waves (sum rebuild, copy into the sampling tree, directional refinement, flattening) is
// independent per spatial leaf; with ~10^4 leaves and ~10^6 quadtree nodes it is spread over the host cores.
int host_threads() {
    static const int n = [] {
        if (const char* e = getenv("PTGPU_HOST_THREADS")) return std::max(1, atoi(e));
        const unsigned hc = std::thread::hardware_concurrency();
        return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
    }();
    return n;
}
template <class F>
void parallel_for(size_t n, size_t grain, F f) {
    const int nt = (int)std::min<size_t>(host_threads(), (n + grain - 1) / std::max<size_t>(1, grain));
    if (nt <= 1) {
        for (size_t i = 0; i < n; i++) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
        pool.emplace_back([&] {
            for (;;) {
                const size_t b = next.fetch_add(grain);
                if (b >= n) return;
                const size_t e = std::min(n, b + grain);
                for (size_t i = b; i < e; i++) f(i);
            }
        });
    for (auto& t : pool) t.join();
}

// DTree::build (dtree.rs:185-198): interior children take the sum of their subtree, bottom-up.
float rebuild_sums(HostDTree& t, uint16_t ni) {
    for (int c = 0; c < 4; c++) {
        const uint16_t ch = t.nodes[ni].child[c];
        if (ch != 0) t.nodes[ni].rad[c] = rebuild_sums(t, ch);
    }
    return t.nodes[ni].sum();
}

// NodeList::free_recursive (dtree.rs:399-410)
void release_subtree(HostDTree& t, uint16_t ni) {
    if (ni == 0) return;
    t.free_slots.push_back(ni);
    const HostDNode n = t.nodes[ni];
    for (int c = 0; c < 4; c++)
        if (n.child[c] != 0) release_subtree(t, n.child[c]);
}

// NodeList::add (dtree.rs:390-398); false when a 16-bit link cannot address the new node.
bool add_node(HostDTree& t, const HostDNode& n, uint16_t* index) {
    if (!t.free_slots.empty()) {
        *index = t.free_slots.back();
        t.free_slots.pop_back();
        t.nodes[*index] = n;
        return true;
    }
    if (t.nodes.size() >= 65535) return false;
    *index = (uint16_t)t.nodes.size();
    t.nodes.push_back(n);
    return true;
}

// DTree::subdivide_and_clear (dtree.rs:291-324): drop subtrees carrying < threshold of the total, split leaves
// carrying more, then zero every sum and the weight.
bool refine_directional(HostDTree& t, float threshold) {
    static thread_local std::vector<std::pair<uint16_t, int>> todo;  // re-used: one heap allocation per thread, not per leaf
    todo.clear();
    todo.emplace_back((uint16_t)0, 1);
    const float total = t.nodes[0].sum();
    while (!todo.empty()) {
        const auto [ni, depth] = todo.back();
        todo.pop_back();
        const HostDNode snapshot = t.nodes[ni];  // the reference works on a copy taken here
        for (int c = 0; c < 4; c++) {
            float frac;
            if (total > 0.0f) frac = snapshot.rad[c] / total;
            else {
                frac = 1.0f;
                for (int k = 0; k < depth; k++) frac *= 0.25f;  // 0.25f32.powi(depth), exact
            }
            if (frac < threshold) {
                release_subtree(t, snapshot.child[c]);
                t.nodes[ni].child[c] = 0;
                continue;
            }
            if (snapshot.child[c] == 0) {
                HostDNode fresh;
                for (int k = 0; k < 4; k++) fresh.rad[k] = snapshot.rad[c] / 4.0f;
                uint16_t idx;
                if (!add_node(t, fresh, &idx)) return false;
                t.nodes[ni].child[c] = idx;
                todo.emplace_back(idx, depth + 1);
            } else {
                todo.emplace_back(snapshot.child[c], depth + 1);
            }
        }
    }
    t.weight = 0.0f;
    for (HostDNode& n : t.nodes)
        for (int k = 0; k < 4; k++) n.rad[k] = 0.0f;
    return true;
}

inline uint2 pack_children(const HostDNode& n) {
    return make_uint2((uint32_t)n.child[0] | ((uint32_t)n.child[1] << 16), (uint32_t)n.child[2] | ((uint32_t)n.child[3] << 16));
}

}  // namespace

void HostSdTree::init(V3 lo, V3 hi) {
    nodes.clear();
    nodes.emplace_back();
    nodes[0].collector.reset_root();
    const Box3 scene{lo, hi};
    bounds = box_around(box_centroid(scene), splat(max_comp(box_extent(scene))));
}

// STree::new_iteration (stree.rs:42-54) and the device image of its result in one go. Per leaf the reference does
// build -> emit_into(sampler) -> subdivide_and_clear(collector); the sampling copy exists only to be read by the next
// wave, so it is written straight into the (page-locked) sampling arena instead of into a host-side DTree first.
bool HostSdTree::refine_and_flatten(SdTreeFlat& out, bool refine, float directional_threshold, float spatial_threshold, uint32_t iteration,
                                    std::string* err) {
    static const bool trace = getenv("PTGPU_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    if (refine) {
        // subdivide_nodes (stree.rs:55-101): threshold = base * sqrt(2^iteration); freshly appended children are
        // visited by the same sweep and may split again. The sweep only decides the topology (it needs nothing but the
        // collector weights): the children of a split are exact copies of their parent's quadtree (collector weight
        // halved), so the copies — most of the work — are left to the parallel pass below; a leaf created here
        // remembers which node still holds the data it descends from.
        float pow2 = 1.0f;
        for (uint32_t k = 0; k < iteration; k++) pow2 *= 2.0f;
        const float threshold = spatial_threshold * sqrtf(pow2);
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].left != 0 || !(nodes[i].collector.weight > threshold)) continue;
            const uint32_t l = (uint32_t)nodes.size(), r = l + 1;
            HostSNode a;
            a.tree_source = nodes[i].tree_source >= 0 ? nodes[i].tree_source : (int64_t)i;
            a.axis = (nodes[i].axis + 1) % 3;
            a.depth = nodes[i].depth + 1;
            a.sampler_weight = nodes[i].sampler_weight;
            a.collector.weight = nodes[i].collector.weight / 2.0f;
            HostSNode b = a;
            nodes[i].left = l;
            nodes[i].right = r;
            nodes.push_back(std::move(a));
            nodes.push_back(std::move(b));
        }
    }
    const double t_split = ms_since(t_begin);

    // leaf numbering (serial, cheap)
    out.snodes.resize(nodes.size());
    out.leaf_snode.clear();
    for (size_t i = 0; i < nodes.size(); i++) {
        const HostSNode& n = nodes[i];
        if (n.left != 0) out.snodes[i] = make_uint4(n.left, n.right, n.axis, 0xffffffffu);
        else {
            out.snodes[i] = make_uint4(0, 0, n.axis, (uint32_t)out.leaf_snode.size());
            out.leaf_snode.push_back((uint32_t)i);
        }
    }
    const size_t n_leaves = out.leaf_snode.size();
    out.leaf_info.resize(n_leaves);

    // pass A (parallel): DTree::build (sums bottom-up); the sampler takes over the collector's weight (emit_into)
    parallel_for(n_leaves, 16, [&](size_t leaf) {
        HostSNode& n = nodes[out.leaf_snode[leaf]];
        if (refine) {
            if (n.tree_source >= 0) {  // split off this iteration: take the ancestor's quadtree (it is interior now, nobody writes it)
                const HostSNode& src = nodes[(size_t)n.tree_source];
                n.collector.nodes = src.collector.nodes;
                n.collector.free_slots = src.collector.free_slots;
            }
            rebuild_sums(n.collector, 0);
            n.sampler_weight = n.collector.weight;  // emit_into (dtree.rs:325-328)
        }
        // DTree::mean() == 0 (dtree.rs:71-77) decides between uniform and tree sampling
        const float mean = n.sampler_weight == 0.0f ? 0.0f : n.collector.nodes[0].sum() / (kFourPi * n.sampler_weight);
        out.leaf_info[leaf] = make_uint4((uint32_t)n.collector.nodes.size(), 0u, mean == 0.0f ? 1u : 0u, 0u);  // .x: size for now
    });
    // sampling-arena offsets; the leaf's sampler word = offset | mean_is_zero << 31 is what Guider::sample needs of the
    // leaf, handed out by the descent itself (leaf node / jump table) so that no leaf_info load follows it
    size_t s_total = 0;
    for (size_t leaf = 0; leaf < n_leaves; leaf++) {
        const uint32_t size = out.leaf_info[leaf].x;
        out.leaf_info[leaf].x = (uint32_t)s_total;
        out.snodes[out.leaf_snode[leaf]].y = (uint32_t)s_total | (out.leaf_info[leaf].z ? 0x80000000u : 0u);
        s_total += size;
    }
    sampler_nodes_total = s_total;
    // top-of-tree jump table
    out.top.resize((size_t)1 << kStreeTopLevels);
    {
        struct Item {
            uint32_t node, prefix, depth;
        };
        std::vector<Item> todo{{0u, 0u, 0u}};
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            const HostSNode& n = nodes[it.node];
            if (n.left != 0 && it.depth < kStreeTopLevels) {
                todo.push_back({n.left, it.prefix << 1, it.depth + 1});
                todo.push_back({n.right, (it.prefix << 1) | 1u, it.depth + 1});
                continue;
            }
            const uint32_t entry = n.left == 0 ? (0x80000000u | (out.snodes[it.node].w << 5) | it.depth) : ((it.node << 5) | it.depth);
            const uint32_t span = 1u << (kStreeTopLevels - it.depth);
            const uint32_t first = it.prefix << (kStreeTopLevels - it.depth);
            const uint2 e2 = make_uint2(entry, n.left == 0 ? out.snodes[it.node].y : 0u);
            for (uint32_t k = 0; k < span; k++) out.top[first + k] = e2;
        }
    }
    out.s_rad.resize(s_total);
    out.s_child.resize(s_total);
    const double t_a = ms_since(t_begin);

    // pass C (parallel): the sampling copy goes into the arena, then DTree::subdivide_and_clear reshapes the collector
    std::atomic<bool> ok{true};
    parallel_for(n_leaves, 16, [&](size_t leaf) {
        HostSNode& n = nodes[out.leaf_snode[leaf]];
        size_t so = out.leaf_info[leaf].x;
        for (const HostDNode& d : n.collector.nodes) {
            out.s_rad[so] = make_float4(d.rad[0], d.rad[1], d.rad[2], d.rad[3]);
            out.s_child[so++] = pack_children(d);
        }
        if (refine && !refine_directional(n.collector, directional_threshold)) ok = false;
    });
    // collector-arena offsets
    size_t c_total = 0;
    for (size_t leaf = 0; leaf < n_leaves; leaf++) {
        out.leaf_info[leaf].y = (uint32_t)c_total;
        c_total += nodes[out.leaf_snode[leaf]].collector.nodes.size();
    }
    out.c_child.resize(c_total);
    out.n_collector_nodes = (uint32_t)c_total;
    // pass E (parallel): collector topology
    parallel_for(n_leaves, 16, [&](size_t leaf) {
        const HostSNode& n = nodes[out.leaf_snode[leaf]];
        size_t co = out.leaf_info[leaf].y;
        for (const HostDNode& d : n.collector.nodes) out.c_child[co++] = pack_children(d);
    });
    if (refine) {
        // interior nodes that served as copy sources give their quadtrees back
        parallel_for(nodes.size(), 64, [&](size_t i) {
            HostSNode& n = nodes[i];
            n.tree_source = -1;
            if (n.left != 0 && !n.collector.nodes.empty()) n.collector = HostDTree{};
        });
    }
    if (trace)
        fprintf(stderr, "[ptgpu]   tree: spatial sweep %.2f ms (serial), sums+offsets %.2f, sampling copy+directional refinement+collector image %.2f\n", t_split,
                t_a - t_split, ms_since(t_begin) - t_a);
    if (!ok) {
        if (err) *err = "a directional quadtree needs more than 65535 nodes (u16 child links)";
        return false;
    }
    return true;
}

void HostSdTree::load_collector_stats(const SdTreeFlat& layout, const float* stats) {
    const float* weights = stats + 4 * (size_t)layout.n_collector_nodes;
    parallel_for(layout.leaf_snode.size(), 16, [&](size_t leaf) {
        HostSNode& n = nodes[layout.leaf_snode[leaf]];
        const float* r = stats + 4 * (size_t)layout.leaf_info[leaf].y;
        for (size_t k = 0; k < n.collector.nodes.size(); k++)
            for (int c = 0; c < 4; c++) n.collector.nodes[k].rad[c] = r[4 * k + c];
        n.collector.weight = weights[leaf];
    });
}

uint64_t HostSdTree::total_sampler_nodes() const { return sampler_nodes_total; }

}  // namespace pt

namespace pt {

void HostSpatialTree::init(V3 lo, V3 hi) {
    nodes.assign(1, SpatialNode{});
    n_leaves = 1;
    const Box3 scene{lo, hi};
    bounds = box_around(box_centroid(scene), splat(max_comp(box_extent(scene))));
}

void HostSpatialTree::next_generation(const float* weights, const uint32_t* sizes, bool refine, float spatial_threshold, uint32_t iteration, SpatialFlat& out) {
    // leaves of the trained generation: take their weights, forget earlier split bookkeeping
    for (SpatialNode& n : nodes)
        if (n.left == 0) {
            n.old_leaf = n.leaf_id;
            n.halvings = 0;
            n.weight = weights ? weights[n.leaf_id] : 0.0f;
        }
    if (refine) {
        // subdivide_nodes (stree.rs:55-101): threshold = base * sqrt(2^iteration); freshly appended children are visited
        // by the same sweep and may split again; both children are copies of the parent with half its weight
        float pow2 = 1.0f;
        for (uint32_t k = 0; k < iteration; k++) pow2 *= 2.0f;
        const float threshold = spatial_threshold * sqrtf(pow2);
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].left != 0 || !(nodes[i].weight > threshold)) continue;
            SpatialNode child;
            child.axis = (nodes[i].axis + 1) % 3;
            child.weight = nodes[i].weight / 2.0f;
            child.old_leaf = nodes[i].old_leaf;
            child.halvings = nodes[i].halvings + 1;
            nodes[i].left = (uint32_t)nodes.size();
            nodes[i].right = nodes[i].left + 1;
            nodes.push_back(child);
            nodes.push_back(child);
        }
    }
    // numbering, sampling-arena offsets (the sampling copy of a leaf is its source's collector as trained)
    out.snodes.resize(nodes.size());
    out.src_leaf.clear();
    out.halvings.clear();
    out.s_off.clear();
    uint64_t s_total = 0;
    for (size_t i = 0; i < nodes.size(); i++) {
        SpatialNode& n = nodes[i];
        if (n.left != 0) {
            out.snodes[i] = make_uint4(n.left, n.right, n.axis, 0xffffffffu);
            continue;
        }
        n.leaf_id = (uint32_t)out.src_leaf.size();
        out.snodes[i] = make_uint4(0, (uint32_t)s_total, n.axis, n.leaf_id);
        out.src_leaf.push_back(n.old_leaf);
        out.halvings.push_back(n.halvings);
        out.s_off.push_back((uint32_t)s_total);
        s_total += sizes ? sizes[n.old_leaf] : 1u;
    }
    n_leaves = (uint32_t)out.src_leaf.size();
    out.sampler_nodes = s_total;
    // top-of-tree jump table (see SdTreeFlat::top)
    out.top.resize((size_t)1 << kStreeTopLevels);
    struct Item {
        uint32_t node, prefix, depth;
    };
    std::vector<Item> todo{{0u, 0u, 0u}};
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        const SpatialNode& n = nodes[it.node];
        if (n.left != 0 && it.depth < kStreeTopLevels) {
            todo.push_back({n.left, it.prefix << 1, it.depth + 1});
            todo.push_back({n.right, (it.prefix << 1) | 1u, it.depth + 1});
            continue;
        }
        const uint32_t entry = n.left == 0 ? (0x80000000u | (n.leaf_id << 5) | it.depth) : ((it.node << 5) | it.depth);
        const uint32_t span = 1u << (kStreeTopLevels - it.depth);
        const uint32_t first = it.prefix << (kStreeTopLevels - it.depth);
        const uint2 e2 = make_uint2(entry, n.left == 0 ? out.snodes[it.node].y : 0u);
        for (uint32_t k = 0; k < span; k++) out.top[first + k] = e2;
    }
}

}  // namespace pt
