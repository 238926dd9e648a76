// Local point order of MapImages' radius samplers, on the host.
//
// The reference samples the points around a camera with an sklearn
// KDTree(leaf_size=50).query_radius (core/data_transform/transforms.py:326,332,
// 380,386), which returns them in the order of the tree's index array; depth
// ties and shared centre pixels of the visibility model are then resolved on
// that order.  This file rebuilds that index array without sklearn:
//
//   n_levels = int(log2(max(1, (n - 1) / leaf_size)) + 1),  n_nodes = 2^n_levels - 1
//   build(node, [a, b)):  leaf when 2 node + 1 >= n_nodes; else
//       dim  = first dimension of largest spread (max - min over the node, float64)
//       nth_element([a, b), a + (b - a) / 2) under "value, then index"
//       build(2 node + 1, [a, mid)), build(2 node + 2, [mid, b))
//
// The comparison is a strict total order, so the two halves are determined as
// sets, but their arrangement -- and with it the order inside the leaves -- is
// whatever the selection algorithm leaves behind.  sklearn's build calls
// libstdc++'s std::nth_element: introselect with a median-of-three pivot
// (first + 1, middle, last - 1, moved to the front), an unguarded Hoare
// partition, a heap select once 2 log2(n) partitions did not narrow the range
// to three elements, and an insertion sort of those.  select_nth below takes the
// same decisions and performs the same swaps:
//
// * It works on (keys, index) records instead of indices into a coordinate
//   table.  The algorithm only compares and moves elements, so the moves are
//   the same, and every comparison stays inside the node's own cache lines.
//   A key is the order-preserving integer image of the float32 coordinate
//   (-0 and +0 share a key, as they compare equal), so "value, then index" is
//   one 64-bit integer comparison.
// * The Hoare partition swaps the k-th element greater than the pivot from the
//   left with the k-th smaller one from the right for as long as the former
//   lies in front of the latter, and ends at first + (number of smaller
//   elements).  That pairing is computed here without data-dependent branches:
//   one pass counts the smaller elements and records one flag per element, a
//   second one lists the misplaced positions on both sides of the end point,
//   and the pairs are swapped in order.  On random coordinates the branches of
//   the textbook loop are mispredicted every other element; this form is about
//   twice as fast and leaves exactly the same arrangement.
//
// The two children of a node are independent: the upper levels of the
// recursion hand their left child to a pool of worker threads that the library
// keeps for the life of the process (creating threads costs more than the build
// itself once the process has mapped a GPU).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mvfusion.h"

namespace mvf { void set_error(const char *fmt, ...); }

namespace {

template <int D> struct Rec { uint32_t k[D]; uint32_t idx; };

// order-preserving image of a float32 (numbers only; -0 -> +0)
inline uint32_t float_key(float f)
{
    f += 0.0f;
    uint32_t b;
    std::memcpy(&b, &f, 4);
    return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
inline float key_float(uint32_t k)
{
    const uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu);
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

template <int D, int DIM> inline uint64_t key_of(const Rec<D> &r) { return ((uint64_t)r.k[DIM] << 32) | r.idx; }

template <int D, int DIM> struct Less {
    bool operator()(const Rec<D> &a, const Rec<D> &b) const { return key_of<D, DIM>(a) < key_of<D, DIM>(b); }
};

// Scratch of one thread: flags and position lists of the partition.  Every thread keeps
// its own for the life of the process: in a process that has mapped a GPU, returning
// large blocks to the system and faulting them in again costs more than the build.
struct Scratch { std::vector<uint8_t> flag; std::vector<uint32_t> left, right; std::vector<uint32_t> recs; };
static Scratch &thread_scratch()
{
    static thread_local Scratch s;
    return s;
}

constexpr int64_t PARTITION_BRANCHLESS_MIN = 96;

// std::__unguarded_partition(lo, hi, pivot): returns the end point.
template <int D, int DIM>
Rec<D> *partition(Rec<D> *lo, Rec<D> *hi, const Rec<D> &pivot_ref, Scratch &S)
{
    const uint64_t pivot = key_of<D, DIM>(pivot_ref);
    const int64_t n = hi - lo;
    if (n < PARTITION_BRANCHLESS_MIN) {
        Rec<D> *first = lo, *last = hi;
        while (true) {
            while (key_of<D, DIM>(*first) < pivot) ++first;
            --last;
            while (pivot < key_of<D, DIM>(*last)) --last;
            if (!(first < last)) return first;
            std::swap(*first, *last);
            ++first;
        }
    }
    if ((int64_t)S.flag.size() < n) { S.flag.resize((size_t)n); S.left.resize((size_t)n / 2 + 1); S.right.resize((size_t)n / 2 + 1); }
    uint8_t *flag = S.flag.data();
    int64_t smaller = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t f = key_of<D, DIM>(lo[i]) < pivot;
        flag[i] = f;
        smaller += f;
    }
    // misplaced: not smaller in front of the end point, smaller behind it (equally many)
    uint32_t *L = S.left.data(), *R = S.right.data();
    int64_t nl = 0, nr = 0;
    for (int64_t i = 0; i < smaller; ++i) { L[nl] = (uint32_t)i; nl += 1 - flag[i]; }
    for (int64_t i = n - 1; i >= smaller; --i) { R[nr] = (uint32_t)i; nr += flag[i]; }
    for (int64_t k = 0; k < nl; ++k) std::swap(lo[L[k]], lo[R[k]]);
    return lo + smaller;
}

// std::nth_element(first, nth, last) of libstdc++ under Less<D, DIM>.
template <int D, int DIM>
void select_nth(Rec<D> *first, Rec<D> *nth, Rec<D> *last, Scratch &S)
{
    if (first == last || nth == last) return;
    const Less<D, DIM> less;
    int64_t depth = 2 * (63 - __builtin_clzll((unsigned long long)(last - first)));
    while (last - first > 3) {
        if (depth == 0) {
            // heap select of the reference's routine itself (never reached on anything but
            // adversarial inputs)
            std::__heap_select(first, nth + 1, last, __gnu_cxx::__ops::__iter_comp_iter(less));
            std::iter_swap(first, nth);
            return;
        }
        --depth;
        Rec<D> *mid = first + (last - first) / 2;
        Rec<D> *a = first + 1, *b = mid, *c = last - 1;
        // std::__move_median_to_first(first, a, b, c)
        if (less(*a, *b)) {
            if (less(*b, *c)) std::iter_swap(first, b);
            else if (less(*a, *c)) std::iter_swap(first, c);
            else std::iter_swap(first, a);
        } else if (less(*a, *c)) std::iter_swap(first, a);
        else if (less(*b, *c)) std::iter_swap(first, c);
        else std::iter_swap(first, b);
        Rec<D> *cut = partition<D, DIM>(first + 1, last, *first, S);
        if (cut <= nth) first = cut;
        else last = cut;
    }
    // std::__insertion_sort on at most three elements
    for (Rec<D> *i = first == last ? last : first + 1; i != last; ++i) {
        const Rec<D> v = *i;
        if (less(v, *first)) {
            std::move_backward(first, i, i + 1);
            *first = v;
        } else {
            Rec<D> *j = i;
            while (less(v, *(j - 1))) { *j = *(j - 1); --j; }
            *j = v;
        }
    }
}

template <int D>
int split_dim(const Rec<D> *r, int64_t n)
{
    int best = 0;
    double best_spread = 0.0;
    for (int j = 0; j < D; ++j) {
        uint32_t hi = r[0].k[j], lo = hi;
        for (int64_t i = 1; i < n; ++i) {
            const uint32_t v = r[i].k[j];
            hi = v > hi ? v : hi;
            lo = v < lo ? v : lo;
        }
        const double spread = (double)key_float(hi) - (double)key_float(lo);
        if (spread > best_spread) { best_spread = spread; best = j; }
    }
    return best;
}

// Worker threads shared by all calls.  A job is a group of tasks; the caller of
// run_group works on tasks itself while it waits for its group.
class Pool {
public:
    struct Group { std::atomic<int> pending{0}; };
    static Pool &get()
    {
        static Pool *p = new Pool();          // never destroyed: workers may outlive static teardown
        return *p;
    }
    int size() const { return (int)workers_.size(); }
    void submit(Group &g, std::function<void()> fn)
    {
        g.pending.fetch_add(1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lock(m_);
            tasks_.emplace_back(Task{&g, std::move(fn)});
        }
        cv_.notify_one();
    }
    // Runs queued tasks (of any group) until group g has none left.
    void wait(Group &g)
    {
        while (g.pending.load(std::memory_order_acquire) > 0) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(m_);
                if (tasks_.empty()) { lock.unlock(); std::this_thread::yield(); continue; }
                t = std::move(tasks_.front());
                tasks_.pop_front();
            }
            t.fn();
            t.group->pending.fetch_sub(1, std::memory_order_release);
        }
    }

private:
    struct Task { Group *group; std::function<void()> fn; };
    Pool()
    {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw ? (int)(hw < 16 ? hw : 16) : 4;
        for (int i = 0; i < n - 1; ++i) {
            workers_.emplace_back([this] { loop(); });
            workers_.back().detach();
        }
    }
    void loop()
    {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this] { return !tasks_.empty(); });
                t = std::move(tasks_.front());
                tasks_.pop_front();
            }
            t.fn();
            t.group->pending.fetch_sub(1, std::memory_order_release);
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> tasks_;
    std::vector<std::thread> workers_;
};

template <int D>
void build(Rec<D> *r, int64_t node, int64_t n_nodes, int64_t n, int spawn_levels, Scratch *S,
           Pool::Group *group)
{
    if (2 * node + 1 >= n_nodes || n < 2) return;
    const int64_t mid = n / 2;
    switch (split_dim<D>(r, n)) {
    case 0: select_nth<D, 0>(r, r + mid, r + n, *S); break;
    case 1: select_nth<D, (D > 1 ? 1 : 0)>(r, r + mid, r + n, *S); break;
    default: select_nth<D, (D > 2 ? 2 : 0)>(r, r + mid, r + n, *S); break;
    }
    if (spawn_levels > 0) {
        Pool::get().submit(*group, [=] {
            build<D>(r, 2 * node + 1, n_nodes, mid, spawn_levels - 1, &thread_scratch(), group);
        });
        build<D>(r + mid, 2 * node + 2, n_nodes, n - mid, spawn_levels - 1, S, group);
    } else {
        build<D>(r, 2 * node + 1, n_nodes, mid, 0, S, group);
        build<D>(r + mid, 2 * node + 2, n_nodes, n - mid, 0, S, group);
    }
}

template <int D>
void order_of(const float *pos, int64_t n, int64_t stride, int64_t leaf_size, int64_t *out, int threads)
{
    Scratch &S = thread_scratch();
    if (S.recs.size() < (size_t)n * (D + 1)) S.recs.resize((size_t)n * (D + 1));
    Rec<D> *recs = reinterpret_cast<Rec<D> *>(S.recs.data());
    static_assert(sizeof(Rec<D>) == 4 * (D + 1), "records are packed 32-bit words");
    for (int64_t i = 0; i < n; ++i) {
        for (int j = 0; j < D; ++j) recs[i].k[j] = float_key(pos[i * stride + j]);
        recs[i].idx = (uint32_t)i;
    }
    const double ratio = (double)(n - 1) / (double)leaf_size;
    const int64_t n_levels = (int64_t)(std::log2(ratio > 1.0 ? ratio : 1.0) + 1.0);
    const int64_t n_nodes = ((int64_t)1 << n_levels) - 1;
    // subtrees handed to the pool: about four per thread, so that uneven ones even out
    int spawn = 0;
    if (threads > 1 && n >= 20000) {
        while ((1 << spawn) < 4 * threads && spawn < 8) ++spawn;
    }
    Pool::Group group;
    build<D>(recs, 0, n_nodes, n, spawn, &S, &group);
    if (spawn > 0) Pool::get().wait(group);
    for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)recs[i].idx;
}

}  // namespace

// h_pos: [n] rows of `stride` floats (host); the first `dims` (2 or 3) columns
// are the tree's coordinates (numbers, no NaN).  h_order: [n] int64, the tree's
// index array.  threads: 1 = the calling thread alone; anything else shares the
// work with the library's worker threads (as many as the host offers, at most 15).
extern "C" int mvf_kdtree_order(
    const float *h_pos, int64_t n, int32_t dims, int64_t stride, int64_t leaf_size,
    int64_t *h_order, int32_t threads)
{
    if (!h_pos || !h_order || n < 1 || n >= ((int64_t)1 << 32) || (dims != 2 && dims != 3) ||
        stride < dims || leaf_size < 1) {
        mvf::set_error("kdtree_order: bad arguments (n=%lld dims=%d stride=%lld leaf_size=%lld)",
                       (long long)n, dims, (long long)stride, (long long)leaf_size);
        return MVF_E_INVALID;
    }
    if (threads < 1) threads = Pool::get().size() + 1;
    try {
        if (dims == 2) order_of<2>(h_pos, n, stride, leaf_size, h_order, threads);
        else order_of<3>(h_pos, n, stride, leaf_size, h_order, threads);
    } catch (const std::exception &e) {
        mvf::set_error("kdtree_order: %s", e.what());
        return MVF_E_LAUNCH;
    }
    return MVF_OK;
}
