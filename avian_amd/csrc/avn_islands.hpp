// avn_islands.hpp -- persistent simulation islands + the bookkeeping of sleeping, host side (include/avian_mi355x.h: avn_island_manager).
//
// What it replaces (paths relative to /root/reference/src): PhysicsIslands (dynamics/solver/islands/mod.rs:404-1280: create / remove,
// add_contact, remove_contact, add_joint, merge_islands, split_island), the island side of the Sleeping set
// (dynamics/solver/islands/sleeping.rs:164-280: awake bits, split candidate, sleep_islands), the SleepIslands / WakeIslands commands
// (:355-540) with ContactGraph::sleep_entity_with / wake_entity_with (collision/contact_types/contact_graph.rs:705-838), and the part of
// the ContactGraph / JointGraph those walk: a collider's / body's edge list in petgraph order (data_structures/stable_graph.rs:640-675).
//
// The reference keeps all of this in intrusive linked lists threaded through ECS components.  Here the SAME ORDERS come out of flat
// arrays: an island owns a vector of its bodies (a merge appends the smaller island's vector, a split writes the depth-first visit
// order), a collider owns two vectors of edge ids in insertion order (the reference's lists are "newest first": walked backwards), the
// slab's vacant keys are a stack.  A contact does not store its island: both of its bodies are in one island from the moment it is
// linked (add_contact merges them; a split keeps bodies joined by a linked contact together), so `island of a contact` is the island of
// whichever of its bodies owns a node.
#pragma once
#include <algorithm>
#include <cstdint>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/avian_mi355x.h"

namespace avn {

class IslandManager {
public:
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    std::string error;

    avn_status body_add(uint32_t body);
    avn_status collider_add(uint32_t collider, uint32_t body);
    avn_status joint_add(uint32_t joint, uint32_t body1, uint32_t body2);
    // remove_joint_from_graph (joint_graph/plugin.rs:163-194): PhysicsIslands::remove_joint (islands/mod.rs:749-812: constraints_removed += 1), the joint out of
    // both bodies' edge lists, and the WakeIslands([island]) it queues for a sleeping island; results as for wake_body
    avn_status joint_remove(uint32_t joint);
    // the host compacted its joint array: joint j -> new_index[j] (NONE: removed, already gone)
    avn_status renumber_joints(const uint32_t* new_index, uint32_t n_old);
    avn_status pair_add(uint32_t contact_id, uint32_t collider1, uint32_t collider2);
    avn_status status_change(uint32_t contact_id, uint32_t flags, uint32_t manifold_count);
    // the closed loop's batches (world/sleeping.hpp): the same calls in the same order, with the next records' cache lines requested ahead
    avn_status pairs_add(const uint32_t* contact_ids, const avn_pair* pairs, uint32_t n);
    avn_status status_changes(const uint32_t* contact_ids, const uint32_t* packed /* flags | manifold count << 16 */, uint32_t n);
    // split_island(candidate) with the contact neighbours handed in as a CSR (round 6): adj[off[b] .. off[b + 1]) = the OTHER body of every contact edge of
    // body b that holds constraint handles and whose other body owns an island node, in the order split_island's walk meets them (the body's colliders in
    // RigidBodyColliders order; per collider outgoing edges newest first, then incoming newest first).  Same result as split_candidate_now(), which derives
    // that order from the manager's own edge lists; here the caller holds it (the closed loop builds it on the device from the rows' insertion stamps).
    avn_status split_candidate_adjacency(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies);
    // ... and when the caller also hands in the COMPONENTS of those edges (+ joints) as a label per body (the closed loop labels them on the device): the split's
    // bookkeeping is then known without the walk -- the pieces are the labels, a piece's key is handed out when the old list first names one of its bodies (the walk
    // starts its pieces in that order: island_remove pushes the old key, every island_insert pops the next), sizes are counts, constraints_removed and the timers
    // start again at 0 -- and only the ORDER inside every piece's body list (the walk's visit order) is outstanding.  The walk runs on a worker thread over the
    // caller's CSR, which must stay untouched until split_join(); until then a piece's list holds its bodies in the OLD list's order.  Every member that reads or
    // reorders the list of a piece of more than one body (SleepIslands / WakeIslands, a merge INTO another island, the next split, despawn, renumbering,
    // avn_islands_state) and every mutation of what the walk reads (joints, island nodes) joins first.  Appending merges do not wait: the walk orders the first n
    // bodies of a list, the newcomers follow them either way.
    avn_status split_candidate_labelled_async(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies, const uint32_t* label);
    // measurement / debugging aid: the CSR rows of the candidate island's bodies against the manager's own edge lists; empty string = equal
    std::string check_adjacency(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies) const;
    avn_status split_join();
    bool split_in_flight() const { return async_.active; }
    ~IslandManager() { if (async_.th.joinable()) async_.th.join(); }
    IslandManager() = default;
    IslandManager(const IslandManager&) = delete;
    void reset() { if (async_.th.joinable()) async_.th.join(); this->~IslandManager(); new (this) IslandManager(); }   // (a fresh manager in place: the worker thread member is not assignable while it runs)
    bool split_pending() const { return candidate_ != NONE && candidate_ < islands_.size() && islands_[candidate_].used && !islands_[candidate_].sleeping && islands_[candidate_].removed != 0; }
    bool has_candidate() const { return candidate_ != NONE; }
    size_t candidate_bodies() const { return candidate_ != NONE && candidate_ < islands_.size() ? islands_[candidate_].bodies.size() : 0; }
    // what the device-built adjacency is keyed by: the rank of every collider in the body-major concatenation of RigidBodyColliders (bodies ascending, a body's
    // colliders in the order they were added); colliders of bodies without a node come after.  collider_epoch() changes whenever the ranks may have.
    uint64_t collider_epoch() const { return col_epoch_; }
    void collider_ranks(const uint32_t* slot_entity, uint32_t n_slots, uint32_t* rank_by_slot);
    avn_status flush_wake();
    avn_status split_candidate_now();
    avn_status sleeping_systems(const float* sleep_timer, const uint8_t* flags, uint32_t n_bodies, float time_to_sleep);
    uint32_t last_flag_awake() const { return last_flag_awake_; }   // bodies whose flags carried bit 2 (owned a SolverBody) in the last sleeping_systems: counted in its one pass
    avn_status wake_body(uint32_t body);
    avn_status sleep_body(uint32_t body);
    // despawn (avn_despawn / avn_islands_collider_remove ...): see the header
    avn_status collider_remove(uint32_t collider);          // the whole of remove_collider, edge order from the manager's own lists
    avn_status remove_collider_edge(uint32_t contact_id);   // one edge of it (the world drives the order from the device's insertion stamps)
    avn_status collider_forget(uint32_t collider);          // ... and the collider's exit from RigidBodyColliders / the node map
    avn_status body_remove(uint32_t body, bool wake);       // BodyIslandNode::on_remove (+ WakeIslands([its island]) when `wake`)
    avn_status wake_island(uint32_t island);
    avn_status renumber_bodies(const uint32_t* new_index, uint32_t n_old);
    const std::vector<uint32_t>& pairs_removed() const { return pairs_removed_; }
    std::vector<uint32_t> collider_edges_in_order(uint32_t collider) const;   // outgoing newest first, then incoming newest first
    avn_status last_result(avn_islands_result* out) const;
    avn_status stats(avn_islands_stats* out) const;
    avn_status state(uint32_t n_bodies, uint32_t* island_of_body, uint32_t* next_in_island, uint8_t* island_sleeping, uint32_t* removed) const;

    // what the world reads directly
    bool body_has_node(uint32_t b) const { return b < node_.size() && node_[b]; }
    bool body_sleeps(uint32_t b) const { return b < asleep_.size() && asleep_[b]; }
    bool has_collider(uint32_t collider) const { const ColRec* r = col_find(collider); return r && r->known; }
    uint32_t island_of(uint32_t b) const { return isl_of_[b]; }
    uint32_t island_key_bound() const { return (uint32_t)islands_.size(); }   // slab keys are below this
    uint32_t last_slept() const { return last_slept_; }
    uint32_t last_woken() const { return last_woken_; }
    const std::vector<uint32_t>& popped() const { return popped_; }
    const std::vector<uint32_t>& pushed() const { return pushed_; }
    const std::vector<uint32_t>& pairs_slept() const { return pairs_slept_; }
    const std::vector<uint32_t>& pairs_woken() const { return pairs_woken_; }
    const std::vector<uint32_t>& bodies_slept() const { return bodies_slept_; }
    const std::vector<uint32_t>& bodies_woken() const { return bodies_woken_; }

private:
    struct Island {
        bool used = false, sleeping = false;
        uint32_t removed = 0;          // constraints_removed
        float timer = 0.0f;            // PhysicsIsland::sleep_timer (only ever combined by merges)
        std::vector<uint32_t> bodies;  // the body list, head first
    };
    struct Contact {
        bool live = false, touching = false, sleeping = false, generates = false, linked = false;
        uint32_t handles = 0;
        uint32_t c1 = NONE, c2 = NONE;       // nodes of the edge lists
        uint32_t b1 = NONE, b2 = NONE;       // bodies of the colliders (NONE: no island node)
        uint32_t rb1 = NONE, rb2 = NONE;     // the colliders' bodies whether they own a node or not (the DFS names the "other" body by entity)
    };
    struct Joint { uint32_t b1 = NONE, b2 = NONE; };
    struct EdgeLists { std::vector<uint32_t> out, in; };   // insertion order; the reference iterates newest first

    std::vector<uint8_t> node_, asleep_;
    uint32_t last_flag_awake_ = 0;
    std::vector<uint32_t> isl_of_;
    std::vector<std::vector<uint32_t>> colliders_of_;
    // colliders by Entity::index(): a dense table for the indices an ECS hands out (small integers), a map behind it for anything above
    struct ColRec { uint32_t body = NONE, node = NONE, rank = NONE; bool known = false; };
    static constexpr uint32_t COL_DENSE = 1u << 24;
    std::vector<ColRec> col_dense_;
    std::unordered_map<uint32_t, ColRec> col_sparse_;
    const ColRec* col_find(uint32_t c) const {
        if (c < COL_DENSE) return c < col_dense_.size() ? &col_dense_[c] : nullptr;
        auto it = col_sparse_.find(c); return it == col_sparse_.end() ? nullptr : &it->second;
    }
    ColRec& col_get(uint32_t c) {
        if (c < COL_DENSE) { if (col_dense_.size() <= c) col_dense_.resize(std::max<size_t>((size_t)c + 1, col_dense_.size() * 2)); return col_dense_[c]; }
        return col_sparse_[c];
    }
    uint32_t col_node(uint32_t c) const { const ColRec* r = col_find(c); return r && r->known ? r->node : NONE; }   // NONE: no edge list yet
    std::vector<EdgeLists> contact_edges_;   // per collider node
    std::vector<EdgeLists> joint_edges_;     // per body
    std::vector<uint8_t> body_has_joint_;    // per body: joint_edges_[b] may be non-empty
    std::vector<Contact> contacts_;
    std::vector<Joint> joints_;
    std::vector<Island> islands_;
    std::vector<uint32_t> vacant_;           // slab: last freed key on top
    uint32_t n_islands_ = 0;
    uint32_t n_sleeping_islands_ = 0, n_nodes_ = 0, n_sleeping_bodies_ = 0;   // kept as they change: avn_sleeping_stats_get is called every frame and used to walk 10^5 bodies
    uint32_t candidate_ = NONE;
    float candidate_timer_ = 0.0f;
    std::vector<uint8_t> awake_;
    std::vector<uint32_t> to_wake_;
    uint32_t merges_ = 0, splits_ = 0, sleeping_pairs_ = 0, last_slept_ = 0, last_woken_ = 0;
    std::vector<uint32_t> popped_, pushed_, pairs_slept_, pairs_woken_, bodies_slept_, bodies_woken_, pairs_removed_;
    // split scratch
    std::vector<uint32_t> mark_contact_, mark_joint_, mark_body_;
    uint32_t mark_gen_ = 0;
    uint64_t col_epoch_ = 1;
    std::vector<uint32_t> split_stack_;
    struct AsyncSplit {
        bool active = false, failed = false;
        struct Piece { uint32_t island, count, first; };
        std::vector<Piece> pieces;                     // in the order the walk starts them
        std::vector<uint32_t> seeds, order, mark, jmark, stack, lab_gen, lab_piece;
        uint32_t gen = 0;
        std::thread th;
        bool holds(uint32_t island) const { for (const Piece& p : pieces) if (p.island == island && p.count > 1) return true; return false; }
    } async_;
    bool async_needs(const std::vector<uint32_t>& ids) const { if (async_.active) for (uint32_t id : ids) if (async_.holds(id)) return true; return false; }
    void async_walk(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies);
    std::vector<std::pair<uint32_t, uint32_t>> split_found_;

    void clear_results();
    uint32_t next_key() const { return vacant_.empty() ? (uint32_t)islands_.size() : vacant_.back(); }
    uint32_t island_insert(Island&& isl);
    void island_remove(uint32_t id);
    uint32_t merge(uint32_t body1, uint32_t body2);
    uint32_t link_contact(uint32_t id);
    uint32_t unlink_contact(uint32_t id);
    uint32_t contact_island(const Contact& c) const { return isl_of_[c.b1 != NONE ? c.b1 : c.b2]; }
    template <class F> void edges_in_reference_order(const EdgeLists& l, uint32_t self_out_node, bool contact_graph, F f) const;
    void sleep_islands(const std::vector<uint32_t>& ids);
    void wake_islands(const std::vector<uint32_t>& ids);
    void split(uint32_t island, const uint32_t* adj_off = nullptr, const uint32_t* adj = nullptr, uint32_t adj_bodies = 0);
    uint32_t node_of(ColRec& r);
};

}  // namespace avn
