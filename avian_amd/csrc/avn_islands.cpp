// avn_islands.cpp -- see avn_islands.hpp.  Host C++ (no device code): the island manager of the closed loop and the avn_islands_* C ABI.
#include <cstring>
#include "avn_islands.hpp"

#include <algorithm>
#include <new>

namespace avn {

void IslandManager::clear_results() { popped_.clear(); pushed_.clear(); pairs_slept_.clear(); pairs_woken_.clear(); bodies_slept_.clear(); bodies_woken_.clear(); pairs_removed_.clear(); }

// slab::Slab::insert / remove: the vacant keys form a stack (remove pushes, insert pops; a fresh key only when the stack is empty)
uint32_t IslandManager::island_insert(Island&& isl) {
    uint32_t key;
    if (vacant_.empty()) { key = (uint32_t)islands_.size(); islands_.emplace_back(); }
    else { key = vacant_.back(); vacant_.pop_back(); }
    isl.used = true;
    islands_[key] = std::move(isl);
    ++n_islands_;
    return key;
}
void IslandManager::island_remove(uint32_t id) {   // PhysicsIslands::remove_island, islands/mod.rs:441-449
    if (candidate_ == id) candidate_ = NONE;
    if (islands_[id].used && islands_[id].sleeping) --n_sleeping_islands_;
    islands_[id] = Island();
    vacant_.push_back(id);
    --n_islands_;
}

avn_status IslandManager::body_add(uint32_t body) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }   // BodyIslandNode::on_add, :1330-1345
    if (node_.size() <= body) { node_.resize((size_t)body + 1, 0); asleep_.resize((size_t)body + 1, 0); isl_of_.resize((size_t)body + 1, NONE); colliders_of_.resize((size_t)body + 1); joint_edges_.resize((size_t)body + 1); }
    if (node_[body]) { error = "islands_body_add: the body already has a node"; return AVN_ERR_STATE; }
    Island isl;
    isl.bodies.push_back(body);
    isl_of_[body] = island_insert(std::move(isl));
    node_[body] = 1;
    ++n_nodes_;
    return AVN_OK;
}
avn_status IslandManager::collider_add(uint32_t collider, uint32_t body) {
    ColRec& r = col_get(collider);
    r.body = body; r.known = true;
    ++col_epoch_;
    if (body != NONE) {
        if (colliders_of_.size() <= body) { colliders_of_.resize((size_t)body + 1); }
        colliders_of_[body].push_back(collider);
    }
    return AVN_OK;
}
uint32_t IslandManager::node_of(ColRec& r) {
    if (r.node != NONE) return r.node;
    r.node = (uint32_t)contact_edges_.size();
    contact_edges_.emplace_back();
    return r.node;
}
avn_status IslandManager::pair_add(uint32_t id, uint32_t c1, uint32_t c2) {   // ContactGraph::add_edge_and_key_with, contact_graph.rs:521-566
    if (contacts_.size() <= id) contacts_.resize((size_t)id + 1);
    if (contacts_[id].live) { error = "islands_pair_add: contact id in use"; return AVN_ERR_STATE; }
    if (!has_collider(c1) || !has_collider(c2)) { error = "islands_pair_add: unknown collider"; return AVN_ERR_BAD_ARG; }
    Contact c;
    c.live = true;
    { ColRec& r1 = col_get(c1); c.c1 = node_of(r1); c.rb1 = r1.body; }
    { ColRec& r2 = col_get(c2); c.c2 = node_of(r2); c.rb2 = r2.body; }
    c.b1 = body_has_node(c.rb1) ? c.rb1 : NONE; c.b2 = body_has_node(c.rb2) ? c.rb2 : NONE;
    contacts_[id] = c;
    contact_edges_[c.c1].out.push_back(id);   // (the reference links at the HEAD of both lists: the walks below run backwards)
    contact_edges_[c.c2].in.push_back(id);
    return AVN_OK;
}

// merge_islands, :814-990: the island with fewer bodies is appended to the other (ties: body1's island stays)
uint32_t IslandManager::merge(uint32_t body1, uint32_t body2) {
    if (!body_has_node(body1)) return isl_of_[body2];
    if (!body_has_node(body2)) return isl_of_[body1];
    uint32_t big = isl_of_[body1], small = isl_of_[body2];
    if (big == small) return big;
    if (islands_[big].bodies.size() < islands_[small].bodies.size()) std::swap(big, small);
    if (async_.active && async_.holds(small)) (void)split_join();   // (its list is appended in list order: the walk's order must be in)
    Island& B = islands_[big]; Island& S = islands_[small];
    for (uint32_t b : S.bodies) isl_of_[b] = big;
    B.bodies.insert(B.bodies.end(), S.bodies.begin(), S.bodies.end());
    B.removed += S.removed;
    if (S.sleeping) { if (!B.sleeping) ++n_sleeping_islands_; B.sleeping = true; B.timer = std::max(S.timer, B.timer); }
    island_remove(small);
    ++merges_;
    return big;
}
uint32_t IslandManager::link_contact(uint32_t id) {   // add_contact, :513-582
    Contact& c = contacts_[id];
    if (c.b1 == NONE && c.b2 == NONE) return NONE;
    const uint32_t isl = merge(c.b1 != NONE ? c.b1 : c.b2, c.b2 != NONE ? c.b2 : c.b1);
    c.linked = true;
    return isl;
}
uint32_t IslandManager::unlink_contact(uint32_t id) {   // remove_contact, :594-660
    Contact& c = contacts_[id];
    c.linked = false;
    const uint32_t isl = contact_island(c);
    islands_[isl].removed += 1;
    return isl;
}
avn_status IslandManager::joint_add(uint32_t jid, uint32_t body1, uint32_t body2) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }   // joint_graph/mod.rs:238-270 + islands/mod.rs:668-735
    if (joints_.size() <= jid) joints_.resize((size_t)jid + 1);
    joints_[jid] = Joint{body1, body2};
    const uint32_t hi = std::max(body1, body2);
    if (joint_edges_.size() <= hi) joint_edges_.resize((size_t)hi + 1);
    joint_edges_[body1].out.push_back(jid);
    joint_edges_[body2].in.push_back(jid);
    if (body_has_joint_.size() <= hi) body_has_joint_.resize((size_t)hi + 1, 0);
    body_has_joint_[body1] = body_has_joint_[body2] = 1;   // (never cleared by joint_remove: "may have joint edges" -- the walk then looks and finds none)
    if (body_has_node(body1) || body_has_node(body2)) merge(body_has_node(body1) ? body1 : body2, body_has_node(body2) ? body2 : body1);
    return AVN_OK;
}

avn_status IslandManager::joint_remove(uint32_t jid) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    clear_results();
    if (jid >= joints_.size() || (joints_[jid].b1 == NONE && joints_[jid].b2 == NONE)) { error = "islands_joint_remove: no such joint"; return AVN_ERR_STATE; }
    const Joint j = joints_[jid];
    // the joint's island is its bodies' island (both are in one island from the moment the joint was added; a split keeps joined bodies together)
    uint32_t isl = NONE;
    if (body_has_node(j.b1)) isl = isl_of_[j.b1]; else if (body_has_node(j.b2)) isl = isl_of_[j.b2];
    if (isl != NONE) islands_[isl].removed += 1;
    auto drop = [&](std::vector<uint32_t>& v) { auto it = std::find(v.rbegin(), v.rend(), jid); if (it != v.rend()) v.erase(std::next(it).base()); };
    if (j.b1 < joint_edges_.size()) drop(joint_edges_[j.b1].out);
    if (j.b2 < joint_edges_.size()) drop(joint_edges_[j.b2].in);
    joints_[jid] = Joint();
    if (isl != NONE && islands_[isl].sleeping) wake_islands({isl});
    return AVN_OK;
}
avn_status IslandManager::renumber_joints(const uint32_t* new_index, uint32_t n_old) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    if (n_old && !new_index) return AVN_ERR_BAD_ARG;
    auto m = [&](uint32_t j) { return j < n_old ? new_index[j] : NONE; };
    std::vector<Joint> nj;
    for (uint32_t j = 0; j < joints_.size() && j < n_old; ++j) {
        if (new_index[j] == NONE) continue;
        if (nj.size() <= new_index[j]) nj.resize((size_t)new_index[j] + 1);
        nj[new_index[j]] = joints_[j];
    }
    joints_.swap(nj);
    for (EdgeLists& l : joint_edges_) { for (uint32_t& e : l.out) e = m(e); for (uint32_t& e : l.in) e = m(e); }
    mark_joint_.clear();
    return AVN_OK;
}

// one iteration of the status loop of NarrowPhase::update, system_param.rs:155-373 (the ConstraintGraph half is the caller's)
avn_status IslandManager::status_change(uint32_t id, uint32_t flags, uint32_t manifold_count) {
    if (id >= contacts_.size() || !contacts_[id].live) { error = "islands_status_change: no such contact"; return AVN_ERR_STATE; }
    Contact& c = contacts_[id];
    const bool generates = flags & AVN_CP_GENERATE_CONSTRAINTS;
    if (flags & AVN_CP_DISJOINT_AABB) {
        if (generates) { c.handles = 0; if (c.linked) unlink_contact(id); }
        if (c.sleeping) --sleeping_pairs_;
        // ContactGraph::remove_edge_by_id: out of both edge lists
        auto drop = [&](std::vector<uint32_t>& v) { auto it = std::find(v.rbegin(), v.rend(), id); if (it != v.rend()) v.erase(std::next(it).base()); };
        drop(contact_edges_[c.c1].out); drop(contact_edges_[c.c2].in);
        contacts_[id] = Contact();
    } else if (flags & AVN_CP_STARTED_TOUCHING) {
        c.touching = true; c.generates = generates;
        if (generates) {
            c.handles = manifold_count;
            const uint32_t isl = link_contact(id);
            if (isl != NONE && islands_[isl].sleeping) to_wake_.push_back(isl);
        }
    } else if (flags & AVN_CP_STOPPED_TOUCHING) {
        c.touching = false; c.generates = generates;
        if (generates && c.handles) {
            c.handles = 0;
            if (c.linked) {   // (a pair whose colliders sit on bodies without island nodes was never linked: link_contact returned NONE)
                const uint32_t isl = unlink_contact(id);
                if (islands_[isl].sleeping) to_wake_.push_back(isl);
            }
        }
    } else if ((flags & AVN_CP_TOUCHING) && (flags & AVN_CP_STARTED_GENERATING_CONSTRAINTS)) {
        c.generates = true;
        c.handles = manifold_count;
        const uint32_t isl = link_contact(id);
        if (isl != NONE && islands_[isl].sleeping) to_wake_.push_back(isl);
    }
    return AVN_OK;
}
avn_status IslandManager::flush_wake() {   // system_param.rs:391-398
    clear_results();
    if (to_wake_.empty()) return AVN_OK;
    std::sort(to_wake_.begin(), to_wake_.end());
    to_wake_.erase(std::unique(to_wake_.begin(), to_wake_.end()), to_wake_.end());
    wake_islands(to_wake_);
    to_wake_.clear();
    return AVN_OK;
}

// EdgeWeights::next: the outgoing list (newest edge first), then the incoming list (newest first)
template <class F> void IslandManager::edges_in_reference_order(const EdgeLists& l, uint32_t, bool, F f) const {
    for (size_t k = l.out.size(); k-- > 0;) f(l.out[k]);
    for (size_t k = l.in.size(); k-- > 0;) f(l.in[k]);
}
// SleepIslands::apply (sleeping.rs:355-420) over ContactGraph::sleep_entity_with (contact_graph.rs:768-838)
void IslandManager::sleep_islands(const std::vector<uint32_t>& ids) {
    if (async_needs(ids)) (void)split_join();
    std::vector<uint32_t> batch;
    for (uint32_t id : ids) {
        if (id >= islands_.size() || !islands_[id].used) continue;
        Island& isl = islands_[id];
        if (isl.sleeping) return;   // (the reference `return`s out of the whole command here)
        isl.sleeping = true; ++n_sleeping_islands_;
        for (uint32_t b : isl.bodies) {
            for (uint32_t col : colliders_of_[b]) {
                const uint32_t nd = col_node(col);
                if (nd == NONE) continue;
                batch.clear();
                edges_in_reference_order(contact_edges_[nd], nd, true, [&](uint32_t e) { if (!contacts_[e].sleeping) batch.push_back(e); });
                for (uint32_t e : batch) {
                    Contact& c = contacts_[e];
                    if (!c.touching) continue;
                    c.sleeping = true; ++sleeping_pairs_;
                    pairs_slept_.push_back(e);
                    if (c.generates) { for (uint32_t k = 0; k < c.handles; ++k) popped_.push_back(e); c.handles = 0; }
                }
            }
            bodies_slept_.push_back(b);
            if (!asleep_[b]) ++n_sleeping_bodies_;
            asleep_[b] = 1;
        }
    }
}
// WakeIslands::apply (:470-540) over wake_entity_with (:705-766)
void IslandManager::wake_islands(const std::vector<uint32_t>& ids) {
    if (async_needs(ids)) (void)split_join();
    std::vector<uint32_t> batch;
    for (uint32_t id : ids) {
        if (id >= islands_.size() || !islands_[id].used || !islands_[id].sleeping) continue;
        Island& isl = islands_[id];
        isl.sleeping = false; --n_sleeping_islands_;
        for (uint32_t b : isl.bodies) {
            for (uint32_t col : colliders_of_[b]) {
                const uint32_t nd = col_node(col);
                if (nd == NONE) continue;
                batch.clear();
                edges_in_reference_order(contact_edges_[nd], nd, true, [&](uint32_t e) { if (contacts_[e].sleeping) batch.push_back(e); });
                for (uint32_t e : batch) {
                    Contact& c = contacts_[e];
                    if (!c.touching) continue;
                    c.sleeping = false; --sleeping_pairs_;
                    pairs_woken_.push_back(e);
                    if (c.generates) { pushed_.push_back(e); c.handles = 1; }   // one manifold per convex pair
                }
            }
            bodies_woken_.push_back(b);
            if (asleep_[b]) --n_sleeping_bodies_;
            asleep_[b] = 0;
        }
    }
}

// split_island, :995-1280: depth-first from the old list's bodies in list order; a body's contacts are collected (collider by collider, edge
// list order, only linked ones with constraint handles not yet claimed) before any of them is claimed, then its joints the same way
void IslandManager::split(uint32_t island, const uint32_t* adj_off, const uint32_t* adj, uint32_t adj_bodies) {
    if (island >= islands_.size() || !islands_[island].used) return;
    if (islands_[island].sleeping || islands_[island].removed == 0) return;
    std::vector<uint32_t> seeds = std::move(islands_[island].bodies);
    island_remove(island);
    ++splits_;
    ++mark_gen_;
    if (mark_body_.size() < node_.size()) mark_body_.resize(node_.size(), 0);
    if (!adj_off && mark_contact_.size() < contacts_.size()) mark_contact_.resize(contacts_.size(), 0);
    if (mark_joint_.size() < joints_.size()) mark_joint_.resize(joints_.size(), 0);
    std::vector<uint32_t>& stack = split_stack_;
    std::vector<std::pair<uint32_t, uint32_t>>& found = split_found_;
    const uint32_t gen = mark_gen_;
    uint32_t* const mark_body = mark_body_.data();
    const uint32_t n_mark = (uint32_t)mark_body_.size();
    const uint8_t* const has_joint = body_has_joint_.data();
    const uint32_t n_has_joint = (uint32_t)std::min(body_has_joint_.size(), joint_edges_.size());
    for (uint32_t seed : seeds) {
        if (mark_body[seed] == gen) continue;
        mark_body[seed] = gen;
        Island isl;
        const uint32_t new_id = next_key();
        stack.assign(1, seed);
        while (!stack.empty()) {
            const uint32_t body = stack.back(); stack.pop_back();
            isl_of_[body] = new_id;
            isl.bodies.push_back(body);
            if (adj_off) {
                // the caller's CSR names, in the walk's order, the other bodies of the edges the walk would collect (handles held, other body owns a node).  The
                // walk's per-contact marks only ever skip an edge whose other body is marked already (the body that claimed the edge), so they are not needed here.
                if (body < adj_bodies) {
                    const uint32_t e1 = adj_off[body + 1];
                    for (uint32_t e = adj_off[body]; e < e1; ++e) {
                        const uint32_t o = adj[e];
                        if (o < n_mark && mark_body[o] != gen) { mark_body[o] = gen; stack.push_back(o); __builtin_prefetch(adj_off + o); }
                    }
                }
            } else {
                found.clear();
                for (uint32_t col : colliders_of_[body]) {
                    const uint32_t nd = col_node(col);
                    if (nd == NONE) continue;
                    edges_in_reference_order(contact_edges_[nd], nd, true, [&](uint32_t e) {
                        const Contact& c = contacts_[e];
                        if (c.linked && mark_contact_[e] == gen) return;
                        if (!c.handles) return;
                        found.push_back({e, c.rb1 == body ? c.rb2 : c.rb1});
                    });
                }
                for (auto& pr : found) {
                    if (body_has_node(pr.second) && mark_body[pr.second] != gen) { stack.push_back(pr.second); mark_body[pr.second] = gen; }
                    mark_contact_[pr.first] = gen;
                }
            }
            if (body < n_has_joint && has_joint[body]) {   // (one byte per body: joint_edges_ is 48 bytes per body, a cache miss per visited body of a jointless pile)
                found.clear();
                edges_in_reference_order(joint_edges_[body], body, false, [&](uint32_t j) {
                    if (mark_joint_[j] == gen) return;
                    found.push_back({j, joints_[j].b1 == body ? joints_[j].b2 : joints_[j].b1});
                });
                for (auto& pr : found) {
                    if (body_has_node(pr.second) && mark_body[pr.second] != gen) { stack.push_back(pr.second); mark_body[pr.second] = gen; }
                    mark_joint_[pr.first] = gen;
                }
            }
        }
        island_insert(std::move(isl));
    }
}
void IslandManager::collider_ranks(const uint32_t* slot_entity, uint32_t n_slots, uint32_t* rank_by_slot) {
    for (ColRec& r : col_dense_) r.rank = NONE;
    for (auto& kv : col_sparse_) kv.second.rank = NONE;
    uint32_t next = 0;
    for (size_t b = 0; b < colliders_of_.size(); ++b)
        for (uint32_t col : colliders_of_[b]) { const ColRec* r = col_find(col); if (r && r->known) col_get(col).rank = next++; }
    for (uint32_t s = 0; s < n_slots; ++s) {
        const ColRec* r = col_find(slot_entity[s]);
        rank_by_slot[s] = (r && r->known && r->rank != NONE) ? r->rank : next++;
    }
}
avn_status IslandManager::split_candidate_now() {
    if (candidate_ == NONE) return AVN_OK;   // (nothing to split: a walk still in flight is left alone)
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    split(candidate_);
    return AVN_OK;
}
avn_status IslandManager::split_candidate_adjacency(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    if (!off || (!adj && n_bodies && off[n_bodies])) { error = "islands_split_candidate_adjacency: null array"; return AVN_ERR_BAD_ARG; }
    if (candidate_ != NONE) split(candidate_, off, adj, n_bodies);
    return AVN_OK;
}
// ---- the split of an island whose pieces are known (labels): bookkeeping now, the order inside the pieces' body lists from a worker thread -----------------
void IslandManager::async_walk(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies) {
    // the walk of split() over the CSR, marks and stack private to the worker; reads seeds, the CSR, the joint lists and the node flags, writes async_.order
    AsyncSplit& a = async_;
    const uint32_t gen = a.gen;
    uint32_t* const mark = a.mark.data();
    const uint32_t n_mark = (uint32_t)a.mark.size();
    const uint8_t* const has_joint = body_has_joint_.data();
    const uint32_t n_has_joint = (uint32_t)std::min(body_has_joint_.size(), joint_edges_.size());
    std::vector<std::pair<uint32_t, uint32_t>> found;
    a.order.clear(); a.order.reserve(a.seeds.size());
    size_t piece = 0;
    bool ok = true;
    for (uint32_t seed : a.seeds) {
        if (seed >= n_mark || mark[seed] == gen) continue;
        mark[seed] = gen;
        if (piece >= a.pieces.size() || a.pieces[piece].first != seed) { ok = false; break; }   // (the labels said otherwise)
        const size_t begin = a.order.size();
        a.stack.assign(1, seed);
        while (!a.stack.empty()) {
            const uint32_t body = a.stack.back(); a.stack.pop_back();
            a.order.push_back(body);
            if (body < n_bodies) {
                const uint32_t e1 = off[body + 1];
                for (uint32_t e = off[body]; e < e1; ++e) {
                    const uint32_t o = adj[e];
                    if (o < n_mark && mark[o] != gen) { mark[o] = gen; a.stack.push_back(o); if (o < n_bodies) __builtin_prefetch(adj + off[o]); }
                }
            }
            if (body < n_has_joint && has_joint[body]) {
                found.clear();
                edges_in_reference_order(joint_edges_[body], body, false, [&](uint32_t j) {
                    if (a.jmark[j] == gen) return;
                    found.push_back({j, joints_[j].b1 == body ? joints_[j].b2 : joints_[j].b1});
                });
                for (auto& pr : found) {
                    if (body_has_node(pr.second) && pr.second < n_mark && mark[pr.second] != gen) { a.stack.push_back(pr.second); mark[pr.second] = gen; }
                    a.jmark[pr.first] = gen;
                }
            }
        }
        if (a.order.size() - begin != a.pieces[piece].count) { ok = false; break; }
        ++piece;
    }
    a.failed = !ok || piece != a.pieces.size() || a.order.size() != a.seeds.size();
}
avn_status IslandManager::split_candidate_labelled_async(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies, const uint32_t* label) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    if (!off || !label || (!adj && n_bodies && off[n_bodies])) { error = "islands_split_candidate_adjacency: null array"; return AVN_ERR_BAD_ARG; }
    if (!split_pending()) return AVN_OK;
    const uint32_t island = candidate_;
    AsyncSplit& a = async_;
    if (++a.gen == 0) { std::fill(a.mark.begin(), a.mark.end(), 0u); std::fill(a.jmark.begin(), a.jmark.end(), 0u); std::fill(a.lab_gen.begin(), a.lab_gen.end(), 0u); a.gen = 1; }
    if (a.mark.size() < node_.size()) a.mark.resize(node_.size(), 0);
    if (a.jmark.size() < joints_.size()) a.jmark.resize(joints_.size(), 0);
    if (a.lab_gen.size() < n_bodies) { a.lab_gen.resize(n_bodies, 0); a.lab_piece.resize(n_bodies, 0); }
    // the pieces, in the order the walk would start them -- the order in which the old list first names a body of the piece -- and their sizes.  First the common
    // case, in one STREAMING pass over the bodies: every member carries the first member's label = still one piece (a settled pile, every other step)
    a.pieces.clear();
    {
        const std::vector<uint32_t>& bs = islands_[island].bodies;
        const uint32_t first = bs.empty() ? NONE : bs[0];
        const uint32_t l0 = first < n_bodies ? label[first] : NONE;
        bool one = l0 < n_bodies, covered = true;
        if (one) {
            size_t members = 0;
            const uint32_t n = std::min<uint32_t>(n_bodies, (uint32_t)node_.size());
            for (uint32_t b = 0; b < n; ++b) members += (size_t)((int)(isl_of_[b] == island) & (int)(node_[b] != 0) & (int)(label[b] == l0));
            one = members == bs.size();
        }
        if (one) a.pieces.push_back({NONE, (uint32_t)bs.size(), first});
        else {
            for (uint32_t b : bs) {   // list order: random reads of the labels
                const uint32_t l = b < n_bodies ? label[b] : NONE;
                if (l >= n_bodies) { covered = false; break; }
                if (a.lab_gen[l] != a.gen) { a.lab_gen[l] = a.gen; a.lab_piece[l] = (uint32_t)a.pieces.size(); a.pieces.push_back({NONE, 0u, b}); }
                ++a.pieces[a.lab_piece[l]].count;
            }
        }
        if (!covered) { a.pieces.clear(); return split_candidate_adjacency(off, adj, n_bodies); }   // (labels do not cover the island: walk now)
    }
    a.seeds = std::move(islands_[island].bodies);
    // split_island's bookkeeping: remove_island (the candidate is cleared, its key goes onto the vacant stack), then one insert per piece
    island_remove(island);
    ++splits_;
    for (AsyncSplit::Piece& p : a.pieces) { Island isl; p.island = island_insert(std::move(isl)); }
    if (a.pieces.size() == 1) {
        // still one piece (a settled pile, every other step): the key comes straight back, nobody changes island -- the list is a copy until the walk's order arrives
        islands_[a.pieces[0].island].bodies = a.seeds;
        if (a.pieces[0].island != island) for (uint32_t b : a.seeds) isl_of_[b] = a.pieces[0].island;
    } else {
        for (AsyncSplit::Piece& p : a.pieces) islands_[p.island].bodies.reserve(p.count);
        for (uint32_t b : a.seeds) {   // until the join: the old list's order inside every piece
            const uint32_t id = a.pieces[a.lab_piece[label[b]]].island;
            islands_[id].bodies.push_back(b);
            isl_of_[b] = id;
        }
    }
    a.active = true; a.failed = false;
    a.th = std::thread([this, off, adj, n_bodies] { async_walk(off, adj, n_bodies); });
    return AVN_OK;
}
std::string IslandManager::check_adjacency(const uint32_t* off, const uint32_t* adj, uint32_t n_bodies) const {
    if (candidate_ == NONE || candidate_ >= islands_.size() || !islands_[candidate_].used) return "";
    std::vector<uint32_t> want;
    for (uint32_t body : islands_[candidate_].bodies) {
        want.clear();
        for (uint32_t col : colliders_of_[body]) {
            const uint32_t nd = col_node(col);
            if (nd == NONE) continue;
            edges_in_reference_order(contact_edges_[nd], nd, true, [&](uint32_t e) {
                const Contact& c = contacts_[e];
                if (!c.handles) return;
                const uint32_t o = c.rb1 == body ? c.rb2 : c.rb1;
                if (body_has_node(o)) want.push_back(o);
            });
        }
        const uint32_t lo = body < n_bodies ? off[body] : 0u, hi = body < n_bodies ? off[body + 1] : 0u;
        bool same = hi - lo == want.size();
        for (uint32_t k = 0; same && k < want.size(); ++k) same = adj[lo + k] == want[k];
        if (!same) {
            std::string w = "body " + std::to_string(body) + ": manager [";
            for (uint32_t x : want) w += std::to_string(x) + " ";
            w += "] device [";
            for (uint32_t k = lo; k < hi && k < lo + 16; ++k) w += std::to_string(adj[k]) + " ";
            return w + "] off " + std::to_string(lo) + ".." + std::to_string(hi);
        }
    }
    return "";
}
avn_status IslandManager::split_join() {
    if (!async_.active) return AVN_OK;
    if (async_.th.joinable()) async_.th.join();
    async_.active = false;
    if (async_.failed) { error = "islands: the labels handed to the split are not the components the walk found (body lists are not the reference's from here on)"; return AVN_ERR_STATE; }
    size_t at = 0;
    for (const AsyncSplit::Piece& p : async_.pieces) {
        if (p.count > 1 && p.island < islands_.size() && islands_[p.island].used && islands_[p.island].bodies.size() >= p.count)
            std::copy(async_.order.begin() + at, async_.order.begin() + at + p.count, islands_[p.island].bodies.begin());
        at += p.count;
    }
    async_.pieces.clear();
    return AVN_OK;
}

// the closed loop's batches: pair_add / status_change over arrays, in array order; the records a call will touch are requested a few calls ahead (a settled
// 100 k-body pile hands over 3*10^4 changes per step scattered over a 45 MB contact table: the loop was one cache miss per change)
avn_status IslandManager::pairs_add(const uint32_t* ids, const avn_pair* pr, uint32_t n) {
    uint32_t hi = 0;
    for (uint32_t i = 0; i < n; ++i) hi = std::max(hi, ids[i]);
    if (n && contacts_.size() <= hi) contacts_.resize((size_t)hi + 1);
    avn_status st;
    // three dependent lines per pair (the contact's record; each collider's list header; the end of each list), requested 12 / 8 / 4 calls ahead
    for (uint32_t i = 0; i < n; ++i) {
        if (i + 12 < n) __builtin_prefetch(&contacts_[ids[i + 12]], 1);
        if (i + 8 < n) {
            const uint32_t n1 = col_node(pr[i + 8].collider1), n2 = col_node(pr[i + 8].collider2);
            if (n1 != NONE) __builtin_prefetch(&contact_edges_[n1], 1);
            if (n2 != NONE) __builtin_prefetch(&contact_edges_[n2], 1);
        }
        if (i + 4 < n) {
            const uint32_t n1 = col_node(pr[i + 4].collider1), n2 = col_node(pr[i + 4].collider2);
            if (n1 != NONE) { const auto& v = contact_edges_[n1].out; if (!v.empty()) __builtin_prefetch(&v.back() + 1, 1); }
            if (n2 != NONE) { const auto& v = contact_edges_[n2].in; if (!v.empty()) __builtin_prefetch(&v.back() + 1, 1); }
        }
        if ((st = pair_add(ids[i], pr[i].collider1, pr[i].collider2)) != AVN_OK) return st;
    }
    return AVN_OK;
}
avn_status IslandManager::status_changes(const uint32_t* cid, const uint32_t* chg, uint32_t n) {
    avn_status st;
    const size_t nc = contacts_.size();
    for (uint32_t k = 0; k < n; ++k) {
        if (k + 12 < n && cid[k + 12] < nc) __builtin_prefetch(&contacts_[cid[k + 12]], 1);
        // a pair that leaves (DISJOINT_AABB) is searched in two edge lists: header 8 calls ahead (the contact's record has arrived by then), data 4 ahead
        if (k + 8 < n && (chg[k + 8] & AVN_CP_DISJOINT_AABB) && cid[k + 8] < nc) {
            const Contact& c = contacts_[cid[k + 8]];
            if (c.live) { __builtin_prefetch(&contact_edges_[c.c1], 1); __builtin_prefetch(&contact_edges_[c.c2], 1); }
        }
        if (k + 4 < n && (chg[k + 4] & AVN_CP_DISJOINT_AABB) && cid[k + 4] < nc) {
            const Contact& c = contacts_[cid[k + 4]];
            if (c.live) {
                const auto& o = contact_edges_[c.c1].out; const auto& i = contact_edges_[c.c2].in;
                if (!o.empty()) __builtin_prefetch(&o.back(), 1);
                if (!i.empty()) __builtin_prefetch(&i.back(), 1);
            }
        }
        if ((st = status_change(cid[k], chg[k] & 0xFFFFu, (chg[k] >> 16) & 0xFFu)) != AVN_OK) return st;
    }
    return AVN_OK;
}

avn_status IslandManager::sleeping_systems(const float* sleep_timer, const uint8_t* flags, uint32_t n_bodies, float time_to_sleep) {
    clear_results();
    if (n_bodies && (!sleep_timer || !flags)) { error = "islands_sleeping_systems: null array"; return AVN_ERR_BAD_ARG; }
    awake_.assign(islands_.size(), 0);
    candidate_timer_ = 0.0f;
    const uint32_t n = std::min<uint32_t>(n_bodies, (uint32_t)node_.size());
    // update_sleeping_states, island side (sleeping.rs:224-239), in body order (the ABI's stand-in for the query's iteration order), and
    // wake_islands_with_sleeping_disabled (:164-182) in the SAME pass (round 6: at 10^5 bodies two passes over the arrays were a fifth of the Sleeping set's host time):
    // the second system only sets awake flags, which the first never reads, so fusing the loops changes no result; the candidate still goes to the FIRST body with the
    // largest timer (strict >, ascending body index)
    const uint8_t* node = node_.data(); const uint32_t* isl_of = isl_of_.data(); uint8_t* awake = awake_.data();
    uint32_t n_flag_awake = 0;
    for (uint32_t b = 0; b < n; ++b) {
        const uint32_t f = flags[b];
        n_flag_awake += (f >> 2) & 1u;
        if (!node[b] || !(f & 3u)) continue;
        const uint32_t isl = isl_of[b];
        if (f & 2u) awake[isl] = 1;
        if (!(f & 1u)) continue;
        const float t = sleep_timer[b];
        if (t < time_to_sleep) awake[isl] = 1;
        else if (t > candidate_timer_ && islands_[isl].removed > 0) { candidate_ = isl; candidate_timer_ = t; }
    }
    for (uint32_t b = n; b < n_bodies; ++b) n_flag_awake += (flags[b] >> 2) & 1u;
    last_flag_awake_ = n_flag_awake;
    // sleep_islands, :243-280, in slab key order
    std::vector<uint32_t> to_sleep, to_wake;
    for (uint32_t k = 0; k < islands_.size(); ++k) {
        const Island& isl = islands_[k];
        if (!isl.used) continue;
        if (awake_[k]) { if (isl.sleeping) to_wake.push_back(k); }
        else if (!isl.sleeping && isl.removed == 0) to_sleep.push_back(k);
    }
    sleep_islands(to_sleep);
    wake_islands(to_wake);
    last_slept_ = (uint32_t)to_sleep.size(); last_woken_ = (uint32_t)to_wake.size();
    return AVN_OK;
}
avn_status IslandManager::wake_body(uint32_t body) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }   // WakeBody, sleeping.rs:438-452
    clear_results();
    if (!body_has_node(body)) { error = "islands_wake_body: the body has no island node"; return AVN_ERR_BAD_ARG; }
    wake_islands({isl_of_[body]});
    return AVN_OK;
}
avn_status IslandManager::sleep_body(uint32_t body) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }   // SleepBody, :296-352
    clear_results();
    if (!body_has_node(body)) { error = "islands_sleep_body: the body has no island node"; return AVN_ERR_BAD_ARG; }
    if (islands_[isl_of_[body]].removed > 0) split(isl_of_[body]);
    sleep_islands({isl_of_[body]});
    return AVN_OK;
}
// ---- despawn ----------------------------------------------------------------------------------------------------------------------------
std::vector<uint32_t> IslandManager::collider_edges_in_order(uint32_t collider) const {
    std::vector<uint32_t> out;
    const uint32_t nd = col_node(collider);
    if (nd == NONE) return out;
    edges_in_reference_order(contact_edges_[nd], nd, true, [&](uint32_t e) { out.push_back(e); });
    return out;
}
// one edge of remove_collider (narrow_phase/mod.rs:411-455 + contact_graph.rs:669-690): a TOUCHING pair that is linked is unlinked, then the edge
// leaves both edge lists
avn_status IslandManager::remove_collider_edge(uint32_t id) {
    if (id >= contacts_.size() || !contacts_[id].live) return AVN_OK;
    Contact& c = contacts_[id];
    if (c.touching && c.linked) unlink_contact(id);
    if (c.sleeping) --sleeping_pairs_;
    auto drop = [&](std::vector<uint32_t>& v) { auto it = std::find(v.rbegin(), v.rend(), id); if (it != v.rend()) v.erase(std::next(it).base()); };
    drop(contact_edges_[c.c1].out); drop(contact_edges_[c.c2].in);
    contacts_[id] = Contact();
    return AVN_OK;
}
avn_status IslandManager::collider_forget(uint32_t collider) {
    if (!has_collider(collider)) return AVN_OK;
    ColRec& r = col_get(collider);
    const uint32_t b = r.body;
    if (b != NONE && b < colliders_of_.size()) { auto& v = colliders_of_[b]; v.erase(std::remove(v.begin(), v.end(), collider), v.end()); }
    ++col_epoch_;
    r = ColRec();   // (its node stays behind, empty; a re-added collider gets a fresh one)
    if (collider >= COL_DENSE) col_sparse_.erase(collider);
    return AVN_OK;
}
avn_status IslandManager::collider_remove(uint32_t collider) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    clear_results();
    if (!has_collider(collider)) { error = "islands_collider_remove: unknown collider"; return AVN_ERR_BAD_ARG; }
    for (uint32_t id : collider_edges_in_order(collider)) {
        const Contact& c = contacts_[id];
        if (c.touching) for (uint32_t k = 0; k < c.handles; ++k) popped_.push_back(id);
        pairs_removed_.push_back(id);
        remove_collider_edge(id);
    }
    return collider_forget(collider);
}
avn_status IslandManager::wake_island(uint32_t island) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    clear_results();
    if (island != NONE) wake_islands({island});
    return AVN_OK;
}
// BodyIslandNode::on_remove, islands/mod.rs:1336-1400: the body leaves its island's list; an island left empty is removed
avn_status IslandManager::body_remove(uint32_t body, bool wake) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    clear_results();
    if (!body_has_node(body)) return AVN_OK;
    const uint32_t island = isl_of_[body];
    Island& I = islands_[island];
    I.bodies.erase(std::remove(I.bodies.begin(), I.bodies.end(), body), I.bodies.end());
    if (I.bodies.empty()) island_remove(island);
    --n_nodes_; if (asleep_[body]) --n_sleeping_bodies_;
    node_[body] = 0; asleep_[body] = 0; isl_of_[body] = NONE;
    if (wake && island < islands_.size() && islands_[island].used) wake_islands({island});
    return AVN_OK;
}
avn_status IslandManager::renumber_bodies(const uint32_t* new_index, uint32_t n_old) {
    { const avn_status js = split_join(); if (js != AVN_OK) return js; }
    if (n_old && !new_index) { error = "islands_renumber_bodies: null map"; return AVN_ERR_BAD_ARG; }
    uint32_t n_new = 0;
    for (uint32_t b = 0; b < n_old; ++b) if (new_index[b] != NONE) n_new = std::max(n_new, new_index[b] + 1u);
    auto m = [&](uint32_t b) { return b != NONE && b < n_old ? new_index[b] : NONE; };
    for (uint32_t b = 0; b < n_old && b < node_.size(); ++b) if (new_index[b] == NONE && node_[b]) { error = "islands_renumber_bodies: a removed body still owns an island node (avn_islands_body_remove first)"; return AVN_ERR_STATE; }
    std::vector<uint8_t> node(n_new, 0), asleep(n_new, 0);
    std::vector<uint32_t> isl_of(n_new, NONE);
    std::vector<std::vector<uint32_t>> cols(n_new);
    std::vector<EdgeLists> jedges(n_new);
    std::vector<uint8_t> hasj(n_new, 0);
    for (uint32_t b = 0; b < n_old && b < node_.size(); ++b) {
        const uint32_t nb = new_index[b];
        if (nb == NONE) continue;
        node[nb] = node_[b]; asleep[nb] = asleep_[b]; isl_of[nb] = isl_of_[b];
        if (b < colliders_of_.size()) cols[nb] = std::move(colliders_of_[b]);
        if (b < joint_edges_.size()) jedges[nb] = std::move(joint_edges_[b]);
        if (b < body_has_joint_.size()) hasj[nb] = body_has_joint_[b];
    }
    node_.swap(node); asleep_.swap(asleep); isl_of_.swap(isl_of); colliders_of_.swap(cols); joint_edges_.swap(jedges); body_has_joint_.swap(hasj);
    ++col_epoch_;
    for (ColRec& r : col_dense_) if (r.known) r.body = m(r.body);
    for (auto& kv : col_sparse_) if (kv.second.known) kv.second.body = m(kv.second.body);
    for (Contact& c : contacts_) if (c.live) { c.b1 = m(c.b1); c.b2 = m(c.b2); c.rb1 = m(c.rb1); c.rb2 = m(c.rb2); }
    for (Joint& j : joints_) { j.b1 = m(j.b1); j.b2 = m(j.b2); }
    for (Island& I : islands_) if (I.used) for (uint32_t& b : I.bodies) b = m(b);
    mark_body_.clear();
    return AVN_OK;
}
avn_status IslandManager::last_result(avn_islands_result* o) const {
    if (!o) return AVN_ERR_BAD_ARG;
    avn_islands_result& r = *o;
    r.popped = popped_.data(); r.n_popped = popped_.size(); r.pushed = pushed_.data(); r.n_pushed = pushed_.size();
    r.pairs_slept = pairs_slept_.data(); r.n_pairs_slept = pairs_slept_.size(); r.pairs_woken = pairs_woken_.data(); r.n_pairs_woken = pairs_woken_.size();
    r.bodies_slept = bodies_slept_.data(); r.n_bodies_slept = bodies_slept_.size(); r.bodies_woken = bodies_woken_.data(); r.n_bodies_woken = bodies_woken_.size();
    r.pairs_removed = pairs_removed_.data(); r.n_pairs_removed = pairs_removed_.size();
    return AVN_OK;
}
avn_status IslandManager::stats(avn_islands_stats* o) const {
    if (!o) return AVN_ERR_BAD_ARG;
    o->n_islands = n_islands_; o->n_sleeping_islands = n_sleeping_islands_; o->n_bodies = n_nodes_; o->n_sleeping_bodies = n_sleeping_bodies_;
    o->merges = merges_; o->splits = splits_; o->split_candidate = candidate_; o->sleeping_pairs = sleeping_pairs_;
    return AVN_OK;
}
avn_status IslandManager::state(uint32_t n_bodies, uint32_t* island_of_body, uint32_t* next_in_island, uint8_t* island_sleeping, uint32_t* removed) const {
    { const avn_status js = const_cast<IslandManager*>(this)->split_join(); if (js != AVN_OK) return js; }
    if (next_in_island) {
        for (uint32_t b = 0; b < n_bodies; ++b) next_in_island[b] = NONE;
        for (const Island& i : islands_)
            if (i.used) for (size_t k = 0; k + 1 < i.bodies.size(); ++k) if (i.bodies[k] < n_bodies) next_in_island[i.bodies[k]] = i.bodies[k + 1];
    }
    for (uint32_t b = 0; b < n_bodies; ++b) {
        const bool n = body_has_node(b);
        if (island_of_body) island_of_body[b] = n ? isl_of_[b] : NONE;
        if (island_sleeping) island_sleeping[b] = n && islands_[isl_of_[b]].sleeping;
        if (removed) removed[b] = n ? islands_[isl_of_[b]].removed : 0u;
    }
    return AVN_OK;
}

}  // namespace avn

// ---- C ABI (include/avian_mi355x.h: avn_islands_*) ----------------------------------------------------------------------------------
struct avn_island_manager { avn::IslandManager m; };
extern "C" {
AVN_API avn_island_manager* avn_islands_create(void) { return new (std::nothrow) avn_island_manager(); }
AVN_API void avn_islands_destroy(avn_island_manager* m) { delete m; }
#define AVN_ISL(call) do { if (!m) return AVN_ERR_BAD_ARG; try { return m->m.call; } catch (...) { m->m.error = "out of host memory"; return AVN_ERR_OOM; } } while (0)
AVN_API avn_status avn_islands_body_add(avn_island_manager* m, uint32_t body) { AVN_ISL(body_add(body)); }
AVN_API avn_status avn_islands_collider_add(avn_island_manager* m, uint32_t collider, uint32_t body) { AVN_ISL(collider_add(collider, body)); }
AVN_API avn_status avn_islands_joint_add(avn_island_manager* m, uint32_t joint, uint32_t b1, uint32_t b2) { AVN_ISL(joint_add(joint, b1, b2)); }
AVN_API avn_status avn_islands_joint_remove(avn_island_manager* m, uint32_t joint) { AVN_ISL(joint_remove(joint)); }
AVN_API avn_status avn_islands_renumber_joints(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old) { AVN_ISL(renumber_joints(new_index, n_old)); }
AVN_API avn_status avn_islands_pair_add(avn_island_manager* m, uint32_t id, uint32_t c1, uint32_t c2) { AVN_ISL(pair_add(id, c1, c2)); }
AVN_API avn_status avn_islands_status_change(avn_island_manager* m, uint32_t id, uint32_t flags, uint32_t manifold_count) { AVN_ISL(status_change(id, flags, manifold_count)); }
AVN_API avn_status avn_islands_flush_wake(avn_island_manager* m) { AVN_ISL(flush_wake()); }
AVN_API avn_status avn_islands_split_candidate(avn_island_manager* m) { AVN_ISL(split_candidate_now()); }
AVN_API avn_status avn_islands_split_candidate_adjacency(avn_island_manager* m, const uint32_t* off, const uint32_t* adj, uint32_t n, const uint32_t* labels) {
    if (!m) return AVN_ERR_BAD_ARG;
    try {
        avn_status st = m->m.split_join();
        if (st != AVN_OK) return st;
        if (labels && off) return m->m.split_candidate_labelled_async(off, adj, n, labels);
        return m->m.split_candidate_adjacency(off, adj, n);
    } catch (...) { m->m.error = "out of host memory"; return AVN_ERR_OOM; }
}
AVN_API avn_status avn_islands_split_join(avn_island_manager* m) { AVN_ISL(split_join()); }
AVN_API avn_status avn_islands_sleeping_systems(avn_island_manager* m, const float* t, const uint8_t* f, uint32_t n, float tts) { AVN_ISL(sleeping_systems(t, f, n, tts)); }
AVN_API avn_status avn_islands_wake_body(avn_island_manager* m, uint32_t body) { AVN_ISL(wake_body(body)); }
AVN_API avn_status avn_islands_sleep_body(avn_island_manager* m, uint32_t body) { AVN_ISL(sleep_body(body)); }
AVN_API avn_status avn_islands_collider_remove(avn_island_manager* m, uint32_t collider) { AVN_ISL(collider_remove(collider)); }
AVN_API avn_status avn_islands_body_remove(avn_island_manager* m, uint32_t body) { AVN_ISL(body_remove(body, true)); }
AVN_API avn_status avn_islands_renumber_bodies(avn_island_manager* m, const uint32_t* new_index, uint32_t n_old) { AVN_ISL(renumber_bodies(new_index, n_old)); }
AVN_API avn_status avn_islands_last_result(avn_island_manager* m, avn_islands_result* out) { AVN_ISL(last_result(out)); }
AVN_API avn_status avn_islands_stats_get(avn_island_manager* m, avn_islands_stats* out) { AVN_ISL(stats(out)); }
AVN_API avn_status avn_islands_state(avn_island_manager* m, uint32_t n_bodies, uint32_t* island_of_body, uint32_t* next_in_island, uint8_t* island_sleeping, uint32_t* removed) {
    AVN_ISL(state(n_bodies, island_of_body, next_in_island, island_sleeping, removed));
}
}
