// Paged radix tree, native host code (include/mi355radix.h).  Mirrors mi355_attn/radix_cache.py - itself pinned to the
// reference's traces (memory/radix_cache.py:73-371) - operation for operation, including the order in which nodes
// are stamped (the stamp order IS the eviction order).  match_prefix stops at a split unless the reference's walk is
// asked for (m355r_set_reference_walk).
#include "../../include/mi355radix.h"

#include <algorithm>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace {

using Key = std::vector<int64_t>;

struct ChunkHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int64_t x : k) { h ^= (uint64_t)x; h *= 1099511628211ull; }
        return (size_t)h;
    }
};

struct Node {
    std::unordered_map<Key, Node*, ChunkHash> children;   // first page_size tokens of the child's key -> child
    Node* parent = nullptr;
    Key key;
    std::vector<int32_t> value;
    int64_t lock_ref = 0;
    uint64_t stamp = 0;
};

struct Tree {
    int32_t ps;
    bool reference_walk = false;                            // match_prefix goes on after a split like the reference (:263-267)
    Node* root = nullptr;
    uint64_t clock = 0;
    int64_t evictable = 0, protected_ = 0, nodes = 0;
    std::vector<int32_t> last_match;                       // pages of the latest match_prefix (m355r_last_match)
    std::unordered_set<const Node*> live;                  // handles that still name a node of this tree

    explicit Tree(int32_t page_size) : ps(page_size) { reset(); }
    ~Tree() { destroy(root); }

    Node* make() { Node* n = new Node(); n->stamp = ++clock; ++nodes; live.insert(n); return n; }   // a new node is stamped (TreeNode.__init__)
    void touch(Node* n) { n->stamp = ++clock; }
    void destroy(Node* n) {
        if (!n) return;
        std::vector<Node*> st{n};
        while (!st.empty()) {
            Node* x = st.back(); st.pop_back();
            for (auto& kv : x->children) st.push_back(kv.second);
            live.erase(x); delete x; --nodes;
        }
    }
    void reset() {
        destroy(root);
        root = make();
        root->lock_ref = 1;
        evictable = protected_ = 0;
    }
    Key chunk(const int64_t* k) const { return Key(k, k + ps); }
    // tokens of a's prefix equal to b's, counted in whole pages (page_size 1: plain common prefix)
    int64_t match_len(const Key& a, const int64_t* b, int64_t nb) const {
        const int64_t lim = std::min<int64_t>((int64_t)a.size(), nb);
        int64_t n = 0;
        while (n + ps <= lim && std::memcmp(a.data() + n, b + n, sizeof(int64_t) * ps) == 0) n += ps;
        return n;
    }
    Node* split(Node* child, int64_t n) {
        Node* top = make();
        top->parent = child->parent;
        top->lock_ref = child->lock_ref;
        top->key.assign(child->key.begin(), child->key.begin() + n);
        top->value.assign(child->value.begin(), child->value.begin() + n / ps);
        top->parent->children[chunk(child->key.data())] = top;          // same first chunk: replaces the child's entry
        child->parent = top;
        child->key.erase(child->key.begin(), child->key.begin() + n);
        child->value.erase(child->value.begin(), child->value.begin() + n / ps);
        top->children[chunk(child->key.data())] = child;
        return top;
    }
};

}  // namespace

extern "C" {

int m355r_version(void) { return 1; }

void* m355r_create(int32_t page_size) { return page_size > 0 ? new Tree(page_size) : nullptr; }
void m355r_destroy(void* t) { delete static_cast<Tree*>(t); }
void m355r_reset(void* t) { if (t) static_cast<Tree*>(t)->reset(); }
void m355r_set_reference_walk(void* t, int32_t on) { if (t) static_cast<Tree*>(t)->reference_walk = on != 0; }

int64_t m355r_match_prefix(void* tp, const int64_t* key, int64_t n, int32_t* out, int64_t cap, int64_t* node_out) {
    Tree* t = static_cast<Tree*>(tp);
    if (!t || n < 0 || (n > 0 && !key) || cap < 0 || (cap > 0 && !out)) return -1;
    if (n == 0) { t->last_match.clear(); if (node_out) *node_out = (int64_t)(intptr_t)t->root; return 0; }    // (no touch: radix_cache.py:119-120)
    n = n / t->ps * t->ps;
    Node* node = t->root;
    t->touch(node);
    t->last_match.clear();
    auto emit = [&](const std::vector<int32_t>& v) { t->last_match.insert(t->last_match.end(), v.begin(), v.end()); };
    const int64_t* k = key;
    int64_t left = n;
    while (left > 0) {
        auto it = node->children.find(t->chunk(k));
        if (it == node->children.end()) break;
        Node* child = it->second;
        t->touch(child);
        const int64_t m = t->match_len(child->key, k, left);
        if (m < (int64_t)child->key.size()) {
            node = t->split(child, m);            // the match ends inside this child: split there and stop
            emit(node->value);
            if (!t->reference_walk) break;        // (reference: goes on from the new upper node WITHOUT consuming the key)
        } else {
            emit(child->value);
            node = child;
            k += m; left -= m;
        }
    }
    if (node_out) *node_out = (int64_t)(intptr_t)node;
    const int64_t written = (int64_t)t->last_match.size();
    if (written > 0 && cap > 0) std::memcpy(out, t->last_match.data(), sizeof(int32_t) * std::min(written, cap));
    return written;
}

int64_t m355r_last_match(void* tp, int32_t* out, int64_t cap) {
    Tree* t = static_cast<Tree*>(tp);
    if (!t || cap < 0 || (cap > 0 && !out)) return -1;
    const int64_t n = (int64_t)t->last_match.size();
    if (n > 0 && cap > 0) std::memcpy(out, t->last_match.data(), sizeof(int32_t) * std::min(n, cap));
    return n;
}

int64_t m355r_insert(void* tp, const int64_t* key, int64_t n, const int32_t* pages, int64_t n_pages) {
    Tree* t = static_cast<Tree*>(tp);
    if (!t || n < 0 || n_pages < 0 || (n > 0 && !key) || (n_pages > 0 && !pages)) return -1;
    const int64_t full = std::min<int64_t>(n / t->ps, n_pages);
    n = full * t->ps;
    Node* node = t->root;
    t->touch(node);
    if (n == 0) return 0;
    const int64_t* k = key;
    const int32_t* v = pages;
    int64_t left = n, matched = 0;
    while (left > 0) {
        auto it = node->children.find(t->chunk(k));
        if (it == node->children.end()) break;
        node = it->second;
        t->touch(node);
        const int64_t m = t->match_len(node->key, k, left);
        matched += m;
        k += m; left -= m; v += m / t->ps;
        if (m < (int64_t)node->key.size()) node = t->split(node, m);
    }
    if (left > 0) {
        Node* leaf = t->make();
        leaf->parent = node;
        leaf->key.assign(k, k + left);
        leaf->value.assign(v, v + left / t->ps);
        node->children[t->chunk(k)] = leaf;
        t->evictable += left;
    }
    return matched;
}

int64_t m355r_inc_lock_ref(void* tp, int64_t h) {
    Tree* t = static_cast<Tree*>(tp);
    Node* node = (Node*)(intptr_t)h;
    if (!t || !node || !t->live.count(node)) return 0;       // a stale handle (its node was evicted) is ignored
    int64_t delta = 0;
    while (node != t->root) {
        if (node->lock_ref == 0) {
            const int64_t n = (int64_t)node->value.size() * t->ps;
            t->evictable -= n; t->protected_ += n; delta -= n;
        }
        ++node->lock_ref;
        node = node->parent;
    }
    return delta;
}

int64_t m355r_dec_lock_ref(void* tp, int64_t h) {
    Tree* t = static_cast<Tree*>(tp);
    Node* node = (Node*)(intptr_t)h;
    if (!t || !node || !t->live.count(node)) return 0;       // a stale handle (its node was evicted) is ignored
    int64_t delta = 0;
    while (node != t->root) {
        if (node->lock_ref == 1) {
            const int64_t n = (int64_t)node->value.size() * t->ps;
            t->evictable += n; t->protected_ -= n; delta += n;
        }
        --node->lock_ref;
        node = node->parent;
    }
    return delta;
}

int64_t m355r_total_size(void* tp) {
    Tree* t = static_cast<Tree*>(tp);
    if (!t) return 0;
    int64_t total = 0;
    std::vector<Node*> st{t->root};
    while (!st.empty()) {
        Node* x = st.back(); st.pop_back();
        total += (int64_t)x->value.size() * t->ps;
        for (auto& kv : x->children) st.push_back(kv.second);
    }
    return total;
}

int64_t m355r_evict(void* tp, int64_t num_pages, int32_t* out, int64_t cap) {
    Tree* t = static_cast<Tree*>(tp);
    if (!t || cap < 0 || (cap > 0 && !out)) return -1;
    if (cap < m355r_total_size(tp) / t->ps) return -1;
    struct Older { bool operator()(const Node* a, const Node* b) const { return a->stamp > b->stamp; } };
    std::priority_queue<Node*, std::vector<Node*>, Older> leaves;
    {
        std::vector<Node*> st{t->root};
        while (!st.empty()) {
            Node* x = st.back(); st.pop_back();
            if (x->children.empty()) leaves.push(x);
            else for (auto& kv : x->children) st.push_back(kv.second);
        }
    }
    int64_t done = 0;
    while (done < num_pages && !leaves.empty()) {
        Node* x = leaves.top(); leaves.pop();
        if (x == t->root) break;
        if (x->lock_ref > 0) continue;
        for (int32_t p : x->value) out[done++] = p;
        Node* parent = x->parent;
        for (auto it = parent->children.begin(); it != parent->children.end(); ++it)
            if (it->second == x) { parent->children.erase(it); break; }
        t->evictable -= (int64_t)x->key.size();
        t->live.erase(x); delete x; --t->nodes;
        if (parent->children.empty()) leaves.push(parent);
    }
    return done;
}

int64_t m355r_evictable_size(void* tp) { return tp ? static_cast<Tree*>(tp)->evictable : 0; }
int64_t m355r_protected_size(void* tp) { return tp ? static_cast<Tree*>(tp)->protected_ : 0; }
int64_t m355r_node_count(void* tp) { return tp ? static_cast<Tree*>(tp)->nodes : 0; }

}  // extern "C"
