// ORACLE (test infrastructure): persistent simulation islands and sleeping, restated from the reference with the reference's own data
// structures -- intrusive doubly linked lists for an island's bodies / contacts / joints, a slab of islands, petgraph-style edge lists.
// Only tests, smoke() and the cpu_baseline leg may use it.  Paths relative to /root/reference/src.
//
//   PhysicsIsland / PhysicsIslands          dynamics/solver/islands/mod.rs:197-505
//   add_contact / remove_contact            :513-582 / :594-660
//   add_joint                               :668-735
//   merge_islands                           :814-990
//   split_island                            :995-1280
//   BodyIslandNode::on_add                  :1330-1345
//   update_sleeping_states (island side)    dynamics/solver/islands/sleeping.rs:184-241
//   wake_islands_with_sleeping_disabled     :164-182
//   sleep_islands                           :243-280
//   SleepBody / SleepIslands                :296-420
//   WakeBody / WakeIslands                  :438-540
//   ContactGraph::sleep_entity_with / wake_entity_with / add_edge_and_key_with / remove_edge_by_id
//                                           collision/contact_types/contact_graph.rs:521-566,599-633,705-838
//   StableUnGraph::add_edge / remove_edge / edge_weights
//                                           data_structures/stable_graph.rs:131-205,286-315,640-675; graph.rs:349-386
//   the status loop                         collision/narrow_phase/system_param.rs:141-398
//   slab 0.4 (third party, pinned in Cargo.lock): `insert` takes the vacant key `next`; `remove(key)` makes `key` the next vacant key
//   (last freed, first reused); iteration is in key order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../include/avian_mi355x.h"

namespace avo {

struct IslandManager {
    static constexpr uint32_t NONE = 0xFFFFFFFFu;
    // IslandNode<Id>, islands/mod.rs:1285-1320
    struct IslandNode { uint32_t island_id = NONE, prev = NONE, next = NONE; bool is_visited = false; };
    struct PhysicsIsland {
        uint32_t id = NONE;
        uint32_t head_body = NONE, tail_body = NONE, body_count = 0;
        uint32_t head_contact = NONE, tail_contact = NONE, contact_count = 0;
        uint32_t head_joint = NONE, tail_joint = NONE, joint_count = 0;
        float sleep_timer = 0.0f;
        bool is_sleeping = false;
        uint32_t constraints_removed = 0;
    };
    // slab::Slab<PhysicsIsland>
    struct Slab {
        struct Entry { bool occupied = false; PhysicsIsland v; uint32_t next_vacant = 0; };
        std::vector<Entry> entries;
        uint32_t next = 0, len = 0;
        uint32_t vacant_key() const { return next; }
        uint32_t insert(const PhysicsIsland& v) {
            const uint32_t key = next;
            ++len;
            if (key == entries.size()) { Entry e; e.occupied = true; e.v = v; entries.push_back(e); next = key + 1; }
            else { next = entries[key].next_vacant; entries[key].occupied = true; entries[key].v = v; }
            return key;
        }
        PhysicsIsland remove(uint32_t key) {
            PhysicsIsland v = entries[key].v;
            entries[key].occupied = false; entries[key].next_vacant = next;
            next = key; --len;
            return v;
        }
        PhysicsIsland* get(uint32_t key) { return key < entries.size() && entries[key].occupied ? &entries[key].v : nullptr; }
    };
    // petgraph adjacency (graph.rs): a node's `next` = heads of its outgoing / incoming edge lists, an edge's `next` = its successors
    struct EdgeLinks { uint32_t node[2] = {NONE, NONE}, next[2] = {NONE, NONE}; bool live = false; };
    struct Lists {
        std::vector<uint32_t> node_next[2];   // [dir][node]
        std::vector<EdgeLinks> edges;
        void ensure_node(uint32_t n) { for (int d = 0; d < 2; ++d) if (node_next[d].size() <= n) node_next[d].resize((size_t)n + 1, NONE); }
        void add_edge(uint32_t e, uint32_t a, uint32_t b) {   // stable_graph.rs:131-205 (a != b)
            if (edges.size() <= e) edges.resize((size_t)e + 1);
            ensure_node(a); ensure_node(b);
            EdgeLinks& ed = edges[e];
            ed.live = true; ed.node[0] = a; ed.node[1] = b;
            ed.next[0] = node_next[0][a]; ed.next[1] = node_next[1][b];
            node_next[0][a] = e; node_next[1][b] = e;
        }
        void remove_edge(uint32_t e) {   // stable_graph.rs:286-315 + graph.rs:349-386 change_edge_links
            EdgeLinks ed = edges[e];
            for (int k = 0; k < 2; ++k) {
                uint32_t& fst = node_next[k][ed.node[k]];
                if (fst == e) fst = ed.next[k];
                else
                    for (uint32_t cur = fst; cur != NONE; cur = edges[cur].next[k])
                        if (edges[cur].next[k] == e) { edges[cur].next[k] = ed.next[k]; break; }
            }
            edges[e] = EdgeLinks();
        }
        // EdgeWeights::next (stable_graph.rs:640-675): the outgoing list, then the incoming list
        template <class F> void for_each_edge(uint32_t a, F f) const {
            if (a >= node_next[0].size()) return;
            for (uint32_t e = node_next[0][a]; e != NONE; e = edges[e].next[0]) f(e);
            for (uint32_t e = node_next[1][a]; e != NONE; e = edges[e].next[1]) { if (edges[e].node[0] == a) continue; f(e); }
        }
        std::vector<uint32_t> edges_of(uint32_t a) const { std::vector<uint32_t> out; for_each_edge(a, [&](uint32_t e) { out.push_back(e); }); return out; }
    };
    // ContactEdge (contact_types/mod.rs) reduced to what islands and sleeping read
    struct ContactEdge {
        bool live = false;
        uint32_t collider1 = NONE, collider2 = NONE;
        uint32_t body1 = NONE, body2 = NONE;     // NONE = the body owns no BodyIslandNode (static / disabled)
        bool touching = false, sleeping = false, generates = false;
        uint32_t handles = 0;                    // constraint_handles.len()
        bool has_island = false;
        IslandNode island;
    };
    struct JointEdge { uint32_t body1 = NONE, body2 = NONE; IslandNode island; bool live = false; };

    std::vector<IslandNode> body_node;            // BodyIslandNode per body
    std::vector<uint8_t> body_has_node, body_sleeping;   // Sleeping component
    std::vector<std::vector<uint32_t>> body_colliders;   // RigidBodyColliders
    std::unordered_map<uint32_t, uint32_t> collider_body;
    std::unordered_map<uint32_t, uint32_t> collider_index;   // entity -> node of the contact graph (dense, first seen first)
    Lists contact_lists, joint_lists;
    std::vector<ContactEdge> contacts;
    std::vector<JointEdge> joints;
    Slab islands;
    uint32_t split_candidate = NONE;
    float split_candidate_sleep_timer = 0.0f;
    std::vector<uint8_t> awake_bits;              // AwakeIslandBitVec
    std::vector<uint32_t> islands_to_wake;
    uint32_t merges = 0, splits = 0, sleeping_pairs = 0;
    // results of the last command batch
    std::vector<uint32_t> popped, pushed, pairs_slept, pairs_woken, bodies_slept, bodies_woken, pairs_removed;
    std::string error;

    void clear_results() { popped.clear(); pushed.clear(); pairs_slept.clear(); pairs_woken.clear(); bodies_slept.clear(); bodies_woken.clear(); pairs_removed.clear(); }
    uint32_t node_of(uint32_t collider) {
        auto it = collider_index.find(collider);
        if (it != collider_index.end()) return it->second;
        uint32_t n = next_node++;   // (never reused: a despawned collider's node stays behind, empty)
        collider_index.emplace(collider, n);
        return n;
    }
    uint32_t next_node = 0;
    bool has_node(uint32_t body) const { return body != NONE && body < body_has_node.size() && body_has_node[body]; }
    bool has_collider(uint32_t collider) const { return collider_body.count(collider) != 0; }

    // BodyIslandNode::on_add, islands/mod.rs:1330-1345
    avn_status body_add(uint32_t body) {
        if (body_node.size() <= body) { body_node.resize((size_t)body + 1); body_has_node.resize((size_t)body + 1, 0); body_sleeping.resize((size_t)body + 1, 0); body_colliders.resize((size_t)body + 1); }
        if (body_has_node[body]) { error = "islands_body_add: the body already has a node"; return AVN_ERR_STATE; }
        PhysicsIsland isl;
        isl.id = islands.vacant_key();
        isl.head_body = body; isl.tail_body = body; isl.body_count = 1;
        const uint32_t id = islands.insert(isl);
        body_node[body] = IslandNode();
        body_node[body].island_id = id;
        body_has_node[body] = 1;
        return AVN_OK;
    }
    avn_status collider_add(uint32_t collider, uint32_t body) {
        collider_body[collider] = body;
        if (body != NONE) {
            if (body_colliders.size() <= body) body_colliders.resize((size_t)body + 1);
            body_colliders[body].push_back(collider);
        }
        return AVN_OK;
    }
    // contact_graph.rs:521-566
    avn_status pair_add(uint32_t id, uint32_t c1, uint32_t c2) {
        if (contacts.size() <= id) contacts.resize((size_t)id + 1);
        if (contacts[id].live) { error = "islands_pair_add: contact id in use"; return AVN_ERR_STATE; }
        auto b1 = collider_body.find(c1), b2 = collider_body.find(c2);
        if (b1 == collider_body.end() || b2 == collider_body.end()) { error = "islands_pair_add: unknown collider"; return AVN_ERR_BAD_ARG; }
        ContactEdge e;
        e.live = true; e.collider1 = c1; e.collider2 = c2;
        e.body1 = has_node(b1->second) ? b1->second : NONE; e.body2 = has_node(b2->second) ? b2->second : NONE;
        contacts[id] = e;
        contact_lists.add_edge(id, node_of(c1), node_of(c2));
        return AVN_OK;
    }

    // ---- despawn (round 4) -------------------------------------------------------------------------------------------------------------
    // the island half of remove_collider's callback for ONE edge (collision/narrow_phase/mod.rs:411-455) followed by the edge's removal
    // (contact_graph.rs:669-690): a TOUCHING pair that is linked into an island is unlinked (constraints_removed += 1)
    void remove_collider_edge(uint32_t id) {
        if (id >= contacts.size() || !contacts[id].live) return;
        ContactEdge& e = contacts[id];
        if (e.touching && e.has_island) remove_contact(id);
        if (e.sleeping) --sleeping_pairs;
        contact_lists.remove_edge(id);
        contacts[id] = ContactEdge();
    }
    // avn_islands_collider_remove: the whole of remove_collider for a host that keeps its own loop -- StableUnGraph::remove_node_with's order (the
    // node's outgoing list from its head, then its incoming list from its head), the handles the host must pop, every edge removed
    avn_status collider_remove_full(uint32_t collider) {
        clear_results();
        if (!collider_body.count(collider)) { error = "islands_collider_remove: unknown collider"; return AVN_ERR_BAD_ARG; }
        auto it = collider_index.find(collider);
        if (it != collider_index.end())
            for (uint32_t id : contact_lists.edges_of(it->second)) {
                const ContactEdge& e = contacts[id];
                if (e.touching) for (uint32_t k = 0; k < e.handles; ++k) popped.push_back(id);
                pairs_removed.push_back(id);
                remove_collider_edge(id);
            }
        collider_remove(collider);
        return AVN_OK;
    }
    avn_status body_remove_and_wake(uint32_t body) {
        clear_results();
        const uint32_t isl = body_remove(body);
        if (isl != NONE) wake_islands({isl});
        return AVN_OK;
    }
    // PhysicsIslands::remove_joint (islands/mod.rs:749-812) + JointGraph::remove_joint (joint_graph/mod.rs:274-285), as remove_joint_from_graph calls them
    // when a joint entity loses its joint component (joint_graph/plugin.rs:163-194): unlink from the island's joint list, constraints_removed += 1,
    // out of both bodies' edge lists; returns the island (NONE: the joint was linked to no island).
    uint32_t joint_remove(uint32_t jid) {
        if (jid >= joints.size() || !joints[jid].live) return NONE;
        const IslandNode ji = joints[jid].island;
        uint32_t island_id = NONE;
        if (ji.island_id != NONE) {
            if (ji.prev != NONE) joints[ji.prev].island.next = ji.next;
            if (ji.next != NONE) joints[ji.next].island.prev = ji.prev;
            if (PhysicsIsland* island = islands.get(ji.island_id)) {
                if (island->head_joint == jid) island->head_joint = ji.next;
                if (island->tail_joint == jid) island->tail_joint = ji.prev;
                island->joint_count -= 1;
                island->constraints_removed += 1;
                island_id = island->id;
            }
        }
        joint_lists.remove_edge(jid);
        joints[jid] = JointEdge();
        return island_id;
    }
    // ... followed by the WakeIslands([island]) the observer queues when that island sleeps (joint_graph/plugin.rs:186-189)
    avn_status joint_remove_and_wake(uint32_t jid) {
        clear_results();
        if (jid >= joints.size() || !joints[jid].live) { error = "islands_joint_remove: no such joint"; return AVN_ERR_STATE; }
        const uint32_t isl = joint_remove(jid);
        if (isl != NONE) wake_islands({isl});
        return AVN_OK;
    }
    // the host compacted its joint array: joint j becomes new_index[j] (NONE = removed, must be gone already); ids are array indices
    void renumber_joints(const std::vector<uint32_t>& new_index) {
        std::vector<JointEdge> nj;
        Lists nl;
        nl.node_next[0] = joint_lists.node_next[0]; nl.node_next[1] = joint_lists.node_next[1];
        auto m = [&](uint32_t j) { return j == NONE ? NONE : new_index[j]; };
        for (int d = 0; d < 2; ++d) for (uint32_t& h : nl.node_next[d]) h = m(h);
        for (uint32_t j = 0; j < joints.size() && j < new_index.size(); ++j) {
            if (new_index[j] == NONE) continue;
            const uint32_t k = new_index[j];
            if (nj.size() <= k) { nj.resize((size_t)k + 1); nl.edges.resize((size_t)k + 1); }
            nj[k] = joints[j];
            nj[k].island.prev = m(nj[k].island.prev); nj[k].island.next = m(nj[k].island.next);
            nl.edges[k] = joint_lists.edges[j];
            nl.edges[k].next[0] = m(nl.edges[k].next[0]); nl.edges[k].next[1] = m(nl.edges[k].next[1]);
        }
        for (uint32_t key = 0; key < islands.entries.size(); ++key)
            if (islands.entries[key].occupied) { PhysicsIsland& i = islands.entries[key].v; i.head_joint = m(i.head_joint); i.tail_joint = m(i.tail_joint); }
        joints.swap(nj); joint_lists = nl;
    }
    // the collider leaves RigidBodyColliders and the ContactGraph's node map (its edges are gone already)
    void collider_remove(uint32_t collider) {
        auto it = collider_body.find(collider);
        if (it == collider_body.end()) return;
        const uint32_t b = it->second;
        if (b != NONE && b < body_colliders.size()) { auto& v = body_colliders[b]; v.erase(std::remove(v.begin(), v.end(), collider), v.end()); }
        collider_body.erase(it);
        collider_index.erase(collider);   // (node indices are not reused: a new collider gets a fresh node)
    }
    // BodyIslandNode::on_remove, islands/mod.rs:1336-1400; returns the island the body was in (NONE: it had no node)
    uint32_t body_remove(uint32_t body) {
        if (!has_node(body)) return NONE;
        const IslandNode bn = body_node[body];
        if (bn.prev != NONE) body_node[bn.prev].next = bn.next;
        if (bn.next != NONE) body_node[bn.next].prev = bn.prev;
        PhysicsIsland* island = islands.get(bn.island_id);
        island->body_count -= 1;
        if (island->head_body == body) {
            island->head_body = bn.next;
            if (island->head_body == NONE) remove_island(bn.island_id);   // the island is empty
        } else if (island->tail_body == body) island->tail_body = bn.prev;
        body_node[body] = IslandNode();
        body_has_node[body] = 0; body_sleeping[body] = 0;
        return bn.island_id;
    }
    // stable compaction of the body indices after a despawn: new_index[old] = new | NONE (removed)
    void renumber_bodies(const std::vector<uint32_t>& new_index, uint32_t n_new) {
        auto m = [&](uint32_t b) { return b != NONE && b < new_index.size() ? new_index[b] : NONE; };
        std::vector<IslandNode> bn(n_new); std::vector<uint8_t> hn(n_new, 0), sl(n_new, 0); std::vector<std::vector<uint32_t>> bc(n_new);
        for (uint32_t b = 0; b < body_node.size() && b < new_index.size(); ++b) {
            const uint32_t nb = new_index[b];
            if (nb == NONE) continue;
            IslandNode x = body_node[b]; x.prev = m(x.prev); x.next = m(x.next);
            bn[nb] = x; hn[nb] = body_has_node[b]; sl[nb] = body_sleeping[b];
            if (b < body_colliders.size()) bc[nb] = body_colliders[b];
        }
        body_node.swap(bn); body_has_node.swap(hn); body_sleeping.swap(sl); body_colliders.swap(bc);
        for (auto& kv : collider_body) kv.second = m(kv.second);
        for (ContactEdge& e : contacts) if (e.live) { e.body1 = m(e.body1); e.body2 = m(e.body2); }
        for (auto& en : islands.entries) if (en.occupied) { en.v.head_body = m(en.v.head_body); en.v.tail_body = m(en.v.tail_body); }
        // the JointGraph's nodes are bodies: its per-node list heads move with them (a despawned body carries no joint: the caller checked)
        for (JointEdge& j : joints) if (j.live) { j.body1 = m(j.body1); j.body2 = m(j.body2); }
        Lists jl;
        for (int d = 0; d < 2; ++d) {
            jl.node_next[d].assign(n_new, NONE);
            for (uint32_t b = 0; b < joint_lists.node_next[d].size() && b < new_index.size(); ++b) if (new_index[b] != NONE) jl.node_next[d][new_index[b]] = joint_lists.node_next[d][b];
        }
        jl.edges = joint_lists.edges;
        for (EdgeLinks& e : jl.edges) if (e.live) { e.node[0] = m(e.node[0]); e.node[1] = m(e.node[1]); }
        joint_lists = jl;
    }

    // merge_islands, islands/mod.rs:814-990
    uint32_t merge_islands(uint32_t body1, uint32_t body2) {
        if (!has_node(body1)) return body_node[body2].island_id;   // (neither: the reference panics; the callers never get there)
        if (!has_node(body2)) return body_node[body1].island_id;
        const uint32_t island_id1 = body_node[body1].island_id, island_id2 = body_node[body2].island_id;
        if (island_id1 == island_id2) return island_id1;
        PhysicsIsland* big = islands.get(island_id1); PhysicsIsland* small = islands.get(island_id2);
        if (big->body_count < small->body_count) std::swap(big, small);
        // 1. remap ids
        for (uint32_t b = small->head_body; b != NONE; b = body_node[b].next) body_node[b].island_id = big->id;
        for (uint32_t c = small->head_contact; c != NONE; c = contacts[c].island.next) contacts[c].island.island_id = big->id;
        for (uint32_t j = small->head_joint; j != NONE; j = joints[j].island.next) joints[j].island.island_id = big->id;
        // 2. append the lists of `small` to `big`
        body_node[big->tail_body].next = small->head_body;
        body_node[small->head_body].prev = big->tail_body;
        big->tail_body = small->tail_body;
        big->body_count += small->body_count;
        if (big->head_contact == NONE) { big->head_contact = small->head_contact; big->tail_contact = small->tail_contact; big->contact_count = small->contact_count; }
        else if (small->head_contact != NONE) {
            contacts[big->tail_contact].island.next = small->head_contact;
            contacts[small->head_contact].island.prev = big->tail_contact;
            big->tail_contact = small->tail_contact;
            big->contact_count += small->contact_count;
        }
        if (big->head_joint == NONE) { big->head_joint = small->head_joint; big->tail_joint = small->tail_joint; big->joint_count = small->joint_count; }
        else if (small->head_joint != NONE) {
            joints[big->tail_joint].island.next = small->head_joint;
            joints[small->head_joint].island.prev = big->tail_joint;
            big->tail_joint = small->tail_joint;
            big->joint_count += small->joint_count;
        }
        big->constraints_removed += small->constraints_removed;
        // 3. sleep state
        if (small->is_sleeping) { big->is_sleeping = true; big->sleep_timer = std::max(small->sleep_timer, big->sleep_timer); }
        // 4. remove the small island
        const uint32_t big_id = big->id, small_id = small->id;
        remove_island(small_id);
        ++merges;
        return big_id;
    }
    void remove_island(uint32_t id) {   // :441-449
        if (split_candidate == id) split_candidate = NONE;
        islands.remove(id);
    }
    // add_contact, :513-582; returns the island (NONE: a body is missing)
    uint32_t add_contact(uint32_t id) {
        ContactEdge& contact = contacts[id];
        // (contact.body1 / body2 are always Some for colliders attached to bodies; what can be missing is the bodies' island NODES)
        if (!has_node(contact.body1) && !has_node(contact.body2)) return NONE;
        const uint32_t island_id = merge_islands(contact.body1 != NONE ? contact.body1 : contact.body2, contact.body2 != NONE ? contact.body2 : contact.body1);
        PhysicsIsland* island = islands.get(island_id);
        IslandNode ci;
        ci.island_id = island->id;
        if (island->head_contact != NONE) { ci.next = island->head_contact; contacts[island->head_contact].island.prev = id; }
        island->head_contact = id;
        if (island->tail_contact == NONE) island->tail_contact = island->head_contact;
        contacts[id].island = ci;
        contacts[id].has_island = true;
        island->contact_count += 1;
        return island_id;
    }
    // remove_contact, :594-660
    uint32_t remove_contact(uint32_t id) {
        ContactEdge& contact = contacts[id];
        const IslandNode ci = contact.island;
        contact.has_island = false; contact.island = IslandNode();
        if (ci.prev != NONE) contacts[ci.prev].island.next = ci.next;
        if (ci.next != NONE) contacts[ci.next].island.prev = ci.prev;
        PhysicsIsland* island = islands.get(ci.island_id);
        if (island->head_contact == id) island->head_contact = ci.next;
        if (island->tail_contact == id) island->tail_contact = ci.prev;
        island->contact_count -= 1;
        island->constraints_removed += 1;
        return ci.island_id;
    }
    // JointGraph::add_joint (joint_graph/mod.rs:238-270) + PhysicsIslands::add_joint (:668-735)
    avn_status joint_add(uint32_t jid, uint32_t body1, uint32_t body2) {
        if (joints.size() <= jid) joints.resize((size_t)jid + 1);
        JointEdge e; e.live = true; e.body1 = body1; e.body2 = body2;
        joints[jid] = e;
        joint_lists.add_edge(jid, body1, body2);
        if (!has_node(body1) && !has_node(body2)) return AVN_OK;
        const uint32_t island_id = merge_islands(has_node(body1) ? body1 : body2, has_node(body2) ? body2 : body1);
        PhysicsIsland* island = islands.get(island_id);
        IslandNode ji;
        ji.island_id = island->id;
        if (island->head_joint != NONE) { ji.next = island->head_joint; joints[island->head_joint].island.prev = jid; }
        island->head_joint = jid;
        if (island->tail_joint == NONE) island->tail_joint = island->head_joint;
        joints[jid].island = ji;
        island->joint_count += 1;
        return AVN_OK;
    }

    // one iteration of the status loop, system_param.rs:155-373 (the ConstraintGraph side lives with the caller)
    avn_status status_change(uint32_t id, uint32_t flags, uint32_t manifold_count) {
        if (id >= contacts.size() || !contacts[id].live) { error = "islands_status_change: no such contact"; return AVN_ERR_STATE; }
        ContactEdge& e = contacts[id];
        const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS;
        if (flags & AVN_CP_DISJOINT_AABB) {
            if (generates) {
                const bool has_island = e.has_island;
                e.handles = 0;
                if (has_island) remove_contact(id);
            }
            // remove_edge_by_id, contact_graph.rs:599-633
            if (e.sleeping) --sleeping_pairs;
            contact_lists.remove_edge(id);
            contacts[id] = ContactEdge();
        } else if (flags & AVN_CP_STARTED_TOUCHING) {
            e.touching = true; e.generates = generates;
            if (generates) {
                e.handles = manifold_count;
                const uint32_t isl = add_contact(id);
                if (isl != NONE && islands.get(isl)->is_sleeping) islands_to_wake.push_back(isl);
            }
        } else if (flags & AVN_CP_STOPPED_TOUCHING) {
            e.touching = false; e.generates = generates;
            if (generates && e.handles) {
                e.handles = 0;
                if (e.has_island) {   // (both bodies without an island node: add_contact returned None, nothing to unlink)
                    const uint32_t isl = remove_contact(id);
                    if (islands.get(isl)->is_sleeping) islands_to_wake.push_back(isl);
                }
            }
        } else if ((flags & AVN_CP_TOUCHING) && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) {
            e.generates = true;
            e.handles = manifold_count;
            const uint32_t isl = add_contact(id);
            if (isl != NONE && islands.get(isl)->is_sleeping) islands_to_wake.push_back(isl);
        }
        return AVN_OK;
    }
    // the deferred WakeIslands of the status loop, system_param.rs:391-398
    avn_status flush_wake() {
        clear_results();
        if (!islands_to_wake.empty()) {
            std::sort(islands_to_wake.begin(), islands_to_wake.end());
            islands_to_wake.erase(std::unique(islands_to_wake.begin(), islands_to_wake.end()), islands_to_wake.end());
            wake_islands(islands_to_wake);
            islands_to_wake.clear();
        }
        return AVN_OK;
    }

    // ContactGraph::sleep_entity_with, contact_graph.rs:768-838, with SleepIslands' callback (sleeping.rs:388-412)
    void sleep_collider(uint32_t collider) {
        auto it = collider_index.find(collider);
        if (it == collider_index.end()) return;
        std::vector<uint32_t> ids;
        contact_lists.for_each_edge(it->second, [&](uint32_t e) { if (!contacts[e].sleeping) ids.push_back(e); });
        for (uint32_t id : ids) {
            ContactEdge& edge = contacts[id];
            if (!edge.touching) continue;
            edge.sleeping = true; ++sleeping_pairs;
            pairs_slept.push_back(id);
            if (edge.touching && edge.generates) {
                for (uint32_t k = 0; k < edge.handles; ++k) popped.push_back(id);
                edge.handles = 0;
            }
        }
    }
    // wake_entity_with, :705-766, with WakeIslands' callback (sleeping.rs:505-524)
    void wake_collider(uint32_t collider) {
        auto it = collider_index.find(collider);
        if (it == collider_index.end()) return;
        std::vector<uint32_t> ids;
        contact_lists.for_each_edge(it->second, [&](uint32_t e) { if (contacts[e].sleeping) ids.push_back(e); });
        for (uint32_t id : ids) {
            ContactEdge& edge = contacts[id];
            if (!edge.touching) continue;
            edge.sleeping = false; --sleeping_pairs;
            pairs_woken.push_back(id);
            if (edge.touching && edge.generates) { pushed.push_back(id); edge.handles = 1; }   // (one manifold per convex pair)
        }
    }
    // SleepIslands::apply, sleeping.rs:355-420
    void sleep_islands(const std::vector<uint32_t>& ids) {
        for (uint32_t island_id : ids) {
            PhysicsIsland* island = islands.get(island_id);
            if (!island) continue;
            if (island->is_sleeping) return;   // (sic: `return`, not `continue`)
            island->is_sleeping = true;
            for (uint32_t b = island->head_body; b != NONE; b = body_node[b].next) {
                for (uint32_t c : body_colliders[b]) sleep_collider(c);
                bodies_slept.push_back(b);
                body_sleeping[b] = 1;
            }
        }
    }
    // WakeIslands::apply, :470-540
    void wake_islands(const std::vector<uint32_t>& ids) {
        for (uint32_t island_id : ids) {
            PhysicsIsland* island = islands.get(island_id);
            if (!island) continue;
            if (!island->is_sleeping) continue;
            island->is_sleeping = false;
            for (uint32_t b = island->head_body; b != NONE; b = body_node[b].next) {
                for (uint32_t c : body_colliders[b]) wake_collider(c);
                bodies_woken.push_back(b);   // sleep_timer.0 = 0.0; Sleeping removed
                body_sleeping[b] = 0;
            }
        }
    }

    // split_island, islands/mod.rs:995-1280
    void split_island(uint32_t island_id) {
        PhysicsIsland* island = islands.get(island_id);
        if (!island) return;
        if (island->is_sleeping) return;
        if (island->constraints_removed == 0) return;
        std::vector<uint32_t> body_ids;
        for (uint32_t b = island->head_body; b != NONE; b = body_node[b].next) { body_ids.push_back(b); body_node[b].is_visited = false; }
        for (uint32_t c = island->head_contact; c != NONE; c = contacts[c].island.next) contacts[c].island.is_visited = false;
        for (uint32_t j = island->head_joint; j != NONE; j = joints[j].island.next) joints[j].island.is_visited = false;
        remove_island(island_id);
        ++splits;
        std::vector<uint32_t> stack;
        for (uint32_t seed : body_ids) {
            if (body_node[seed].is_visited) continue;
            body_node[seed].is_visited = true;
            PhysicsIsland isl;
            const uint32_t new_id = islands.vacant_key();
            isl.id = new_id;
            stack.push_back(seed);
            while (!stack.empty()) {
                const uint32_t body = stack.back(); stack.pop_back();
                IslandNode& bn = body_node[body];
                bn.island_id = new_id;
                if (isl.tail_body != NONE) body_node[isl.tail_body].next = body;
                bn.prev = isl.tail_body; bn.next = NONE;
                isl.tail_body = body;
                if (isl.head_body == NONE) isl.head_body = body;
                isl.body_count += 1;
                // the contacts of the body, collected first (:1112-1148)
                std::vector<std::pair<uint32_t, uint32_t>> contact_edges;
                for (uint32_t collider : body_colliders[body]) {
                    auto it = collider_index.find(collider);
                    if (it == collider_index.end()) continue;
                    contact_lists.for_each_edge(it->second, [&](uint32_t e) {
                        const ContactEdge& ce = contacts[e];
                        if (ce.has_island && ce.island.is_visited) return;
                        if (ce.handles == 0) return;
                        // (body1 / body2 are Some for every collider of a body; `other` may be a body without a node)
                        const uint32_t b1 = collider_body[ce.collider1], b2 = collider_body[ce.collider2];
                        contact_edges.push_back({e, b1 == body ? b2 : b1});
                    });
                }
                for (auto& pr : contact_edges) {
                    const uint32_t cid = pr.first, other = pr.second;
                    if (has_node(other) && !body_node[other].is_visited) { stack.push_back(other); body_node[other].is_visited = true; }
                    if (isl.tail_contact != NONE) contacts[isl.tail_contact].island.next = cid;
                    IslandNode& ci = contacts[cid].island;
                    ci.is_visited = true; ci.island_id = new_id; ci.prev = isl.tail_contact; ci.next = NONE;
                    isl.tail_contact = cid;
                    if (isl.head_contact == NONE) isl.head_contact = cid;
                    isl.contact_count += 1;
                }
                std::vector<std::pair<uint32_t, uint32_t>> joint_edges;
                joint_lists.for_each_edge(body, [&](uint32_t j) {
                    if (joints[j].island.is_visited) return;
                    joint_edges.push_back({j, joints[j].body1 == body ? joints[j].body2 : joints[j].body1});
                });
                for (auto& pr : joint_edges) {
                    const uint32_t jid = pr.first, other = pr.second;
                    if (has_node(other) && !body_node[other].is_visited) { stack.push_back(other); body_node[other].is_visited = true; }
                    if (isl.tail_joint != NONE) joints[isl.tail_joint].island.next = jid;
                    IslandNode& ji = joints[jid].island;
                    ji.is_visited = true; ji.island_id = new_id; ji.prev = isl.tail_joint; ji.next = NONE;
                    isl.tail_joint = jid;
                    if (isl.head_joint == NONE) isl.head_joint = jid;
                    isl.joint_count += 1;
                }
            }
            islands.insert(isl);
        }
    }
    avn_status split_candidate_now() { if (split_candidate != NONE) split_island(split_candidate); return AVN_OK; }   // split_island system, :160-178
    // avn_islands_split_candidate_adjacency on the oracle: the caller's CSR is CHECKED -- for every body of the candidate island the row must name, in order, the other
    // bodies of the edges split_island's walk would collect from the petgraph lists (:1112-1148) that own a node -- and then the walk above runs on the oracle's own lists
    avn_status split_candidate_adjacency(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies) {
        if (!off) return AVN_ERR_BAD_ARG;
        if (split_candidate == NONE) return AVN_OK;
        PhysicsIsland* island = islands.get(split_candidate);
        if (island && !island->is_sleeping && island->constraints_removed != 0) {
            for (uint32_t body = island->head_body; body != NONE; body = body_node[body].next) {
                std::vector<uint32_t> want;
                for (uint32_t collider : body_colliders[body]) {
                    auto it = collider_index.find(collider);
                    if (it == collider_index.end()) continue;
                    contact_lists.for_each_edge(it->second, [&](uint32_t e) {
                        const ContactEdge& ce = contacts[e];
                        if (ce.handles == 0) return;
                        const uint32_t b1 = collider_body[ce.collider1], b2 = collider_body[ce.collider2];
                        const uint32_t other = b1 == body ? b2 : b1;
                        if (has_node(other)) want.push_back(other);
                    });
                }
                const uint32_t lo = body < n_bodies ? off[body] : 0u, hi = body < n_bodies ? off[body + 1] : 0u;
                bool same = hi - lo == want.size();
                for (uint32_t k = 0; same && k < want.size(); ++k) same = adj[lo + k] == want[k];
                if (!same) { error = "islands_split_candidate_adjacency: the adjacency row of body " + std::to_string(body) + " is not the walk's edge order"; return AVN_ERR_STATE; }
            }
        }
        split_island(split_candidate);
        return AVN_OK;
    }

    // the Sleeping set: island side of update_sleeping_states (sleeping.rs:224-239), wake_islands_with_sleeping_disabled (:164-182),
    // sleep_islands (:243-280), then the two queued commands
    avn_status sleeping_systems(const float* sleep_timer, const uint8_t* flags, uint32_t n_bodies, float time_to_sleep) {
        clear_results();
        if (awake_bits.size() < islands.entries.size()) awake_bits.resize(islands.entries.size(), 0);
        split_candidate_sleep_timer = 0.0f;
        for (uint32_t b = 0; b < n_bodies && b < body_has_node.size(); ++b) {
            if (!body_has_node[b] || !(flags[b] & 1u)) continue;
            const uint32_t isl = body_node[b].island_id;
            if (sleep_timer[b] < time_to_sleep) awake_bits[isl] = 1;
            else if (PhysicsIsland* island = islands.get(isl)) {
                if (island->constraints_removed > 0 && sleep_timer[b] > split_candidate_sleep_timer) { split_candidate = isl; split_candidate_sleep_timer = sleep_timer[b]; }
            }
        }
        for (uint32_t b = 0; b < n_bodies && b < body_has_node.size(); ++b)
            if (body_has_node[b] && (flags[b] & 2u)) awake_bits[body_node[b].island_id] = 1;
        std::vector<uint32_t> sleep_buffer, wake_buffer;
        for (uint32_t k = 0; k < islands.entries.size(); ++k) {
            if (!islands.entries[k].occupied) continue;
            PhysicsIsland& island = islands.entries[k].v;
            if (awake_bits[island.id]) { if (island.is_sleeping) wake_buffer.push_back(island.id); }
            else if (!island.is_sleeping && island.constraints_removed == 0) sleep_buffer.push_back(island.id);
        }
        sleep_islands(sleep_buffer);
        wake_islands(wake_buffer);
        awake_bits.assign(islands.entries.size(), 0);   // set_bit_count_and_clear(islands.len()) (the bits are indexed by id: cleared up to the slab's capacity)
        last_slept = (uint32_t)sleep_buffer.size(); last_woken = (uint32_t)wake_buffer.size();
        return AVN_OK;
    }
    uint32_t last_slept = 0, last_woken = 0;
    // WakeBody / SleepBody, sleeping.rs:283-352,438-452
    avn_status wake_body(uint32_t body) {
        clear_results();
        if (!has_node(body)) { error = "islands_wake_body: the body has no island node"; return AVN_ERR_BAD_ARG; }
        wake_islands({body_node[body].island_id});
        return AVN_OK;
    }
    avn_status sleep_body(uint32_t body) {
        clear_results();
        if (!has_node(body)) { error = "islands_sleep_body: the body has no island node"; return AVN_ERR_BAD_ARG; }
        uint32_t isl = body_node[body].island_id;
        if (PhysicsIsland* island = islands.get(isl)) if (island->constraints_removed > 0) split_island(isl);
        isl = body_node[body].island_id;
        sleep_islands({isl});
        return AVN_OK;
    }
    avn_status stats(avn_islands_stats* o) {
        if (!o) return AVN_ERR_BAD_ARG;
        uint32_t ns = 0, nb = 0, nsb = 0;
        for (auto& e : islands.entries) if (e.occupied && e.v.is_sleeping) ++ns;
        for (size_t b = 0; b < body_has_node.size(); ++b) if (body_has_node[b]) { ++nb; if (body_sleeping[b]) ++nsb; }
        o->n_islands = islands.len; o->n_sleeping_islands = ns; o->n_bodies = nb; o->n_sleeping_bodies = nsb;
        o->merges = merges; o->splits = splits; o->split_candidate = split_candidate; o->sleeping_pairs = sleeping_pairs;
        return AVN_OK;
    }
    avn_status state(uint32_t n_bodies, uint32_t* island_of_body, uint32_t* next_in_island, uint8_t* island_sleeping, uint32_t* removed) {
        for (uint32_t b = 0; b < n_bodies; ++b) {
            const bool n = b < body_has_node.size() && body_has_node[b];
            PhysicsIsland* isl = n ? islands.get(body_node[b].island_id) : nullptr;
            if (island_of_body) island_of_body[b] = n ? body_node[b].island_id : NONE;
            if (next_in_island) next_in_island[b] = n ? body_node[b].next : NONE;
            if (island_sleeping) island_sleeping[b] = isl && isl->is_sleeping;
            if (removed) removed[b] = isl ? isl->constraints_removed : 0u;
        }
        return AVN_OK;
    }
    // islands/mod.rs:255-400 validate(): the linked lists of every island are consistent with the nodes' island ids and the counts
    bool validate(std::string& why) {
        std::vector<uint32_t> seen_body(body_node.size(), 0);
        for (auto& en : islands.entries) {
            if (!en.occupied) continue;
            const PhysicsIsland& isl = en.v;
            uint32_t count = 0, prev = NONE;
            for (uint32_t b = isl.head_body; b != NONE; b = body_node[b].next) {
                if (body_node[b].island_id != isl.id) { why = "body with a foreign island id"; return false; }
                if (body_node[b].prev != prev) { why = "broken prev link"; return false; }
                if (seen_body[b]++) { why = "body in two lists"; return false; }
                prev = b; ++count;
                if (count > body_node.size()) { why = "cycle"; return false; }
            }
            if (count != isl.body_count || prev != isl.tail_body || count == 0) { why = "body count / tail mismatch"; return false; }
            count = 0; prev = NONE;
            for (uint32_t c = isl.head_contact; c != NONE; c = contacts[c].island.next) {
                if (!contacts[c].live || !contacts[c].has_island || contacts[c].island.island_id != isl.id || contacts[c].island.prev != prev) { why = "contact list inconsistent"; return false; }
                prev = c; ++count;
                if (count > contacts.size()) { why = "contact cycle"; return false; }
            }
            if (count != isl.contact_count || prev != isl.tail_contact) { why = "contact count / tail mismatch"; return false; }
        }
        for (size_t b = 0; b < body_node.size(); ++b) if (body_has_node[b] && seen_body[b] != 1) { why = "a body with a node is in no island list"; return false; }
        return true;
    }
};

}  // namespace avo
