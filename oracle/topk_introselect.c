/* TEST INFRASTRUCTURE -- C restatement of the index selection behind
 *   torch.topk(score, k, dim=-1, largest=False)   (CPU)
 * as called by the reference at model/futr_safuser_tokenfusion.py:53-54.
 *
 * Not in /root/reference: the algorithm lives in PyTorch's ATen CPU kernel
 * (aten/src/ATen/native/cpu/TopKImpl.h, topk_impl_loop; torch 2.10.0 in this image).  For
 * k*64 > n (always true here: k = C/4) it fills a queue of (value, index) pairs and runs
 *   std::nth_element(queue, queue + k - 1, queue + n, cmp),
 *   cmp(x, y) = (!isnan(x.v) && isnan(y.v)) || (x.v < y.v)
 * then (sorted=true) sorts the first k-1 entries -- which changes their order, not the SET.
 * std::nth_element is libstdc++'s introselect (bits/stl_algo.h: __introselect,
 * __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition, __heap_select,
 * __insertion_sort); this file restates that published algorithm so ties (in train mode every
 * score is equal) resolve exactly as they do in the reference.
 * Pinned against torch.topk itself in tests/test_oracle_topk.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { float v; int64_t i; } elem_t;

static inline int lt(const elem_t* x, const elem_t* y) {
    return ((!isnan(x->v)) && isnan(y->v)) || (x->v < y->v);
}
static inline void swp(elem_t* a, elem_t* b) { elem_t t = *a; *a = *b; *b = t; }

static void move_median_to_first(elem_t* result, elem_t* a, elem_t* b, elem_t* c) {
    if (lt(a, b)) {
        if (lt(b, c)) swp(result, b);
        else if (lt(a, c)) swp(result, c);
        else swp(result, a);
    } else if (lt(a, c)) swp(result, a);
    else if (lt(b, c)) swp(result, c);
    else swp(result, b);
}

static elem_t* unguarded_partition(elem_t* first, elem_t* last, elem_t* pivot) {
    for (;;) {
        while (lt(first, pivot)) ++first;
        --last;
        while (lt(pivot, last)) --last;
        if (!(first < last)) return first;
        swp(first, last);
        ++first;
    }
}

static void push_heap_(elem_t* first, int64_t hole, int64_t top, elem_t value) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && lt(first + parent, &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

static void adjust_heap(elem_t* first, int64_t hole, int64_t len, elem_t value) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (lt(first + child, first + (child - 1))) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    push_heap_(first, hole, top, value);
}

static void make_heap_(elem_t* first, elem_t* last) {
    const int64_t len = last - first;
    if (len < 2) return;
    int64_t parent = (len - 2) / 2;
    for (;;) {
        elem_t value = first[parent];
        adjust_heap(first, parent, len, value);
        if (parent == 0) return;
        parent--;
    }
}

static void heap_select(elem_t* first, elem_t* middle, elem_t* last) {
    make_heap_(first, middle);
    for (elem_t* i = middle; i < last; ++i)
        if (lt(i, first)) {               /* __pop_heap(first, middle, i) */
            elem_t value = *i;
            *i = *first;
            adjust_heap(first, 0, middle - first, value);
        }
}

static void insertion_sort_(elem_t* first, elem_t* last) {
    if (first == last) return;
    for (elem_t* i = first + 1; i != last; ++i) {
        if (lt(i, first)) {
            elem_t val = *i;
            for (elem_t* j = i; j != first; --j) *j = *(j - 1);
            *first = val;
        } else {                           /* __unguarded_linear_insert */
            elem_t val = *i;
            elem_t* next = i - 1;
            elem_t* cur = i;
            while (lt(&val, next)) { *cur = *next; cur = next; --next; }
            *cur = val;
        }
    }
}

static int64_t lg2(int64_t n) { int64_t r = 0; while (n > 1) { n >>= 1; ++r; } return r; }

static void introselect(elem_t* first, elem_t* nth, elem_t* last, int64_t depth_limit) {
    while (last - first > 3) {
        if (depth_limit == 0) {
            heap_select(first, nth + 1, last);
            swp(first, nth);
            return;
        }
        --depth_limit;
        elem_t* mid = first + (last - first) / 2;
        move_median_to_first(first, first + 1, mid, last - 1);
        elem_t* cut = unguarded_partition(first + 1, last, first);
        if (cut <= nth) first = cut; else last = cut;
    }
    insertion_sort_(first, last);
}

/* out[0..k) = indices torch.topk(score[0..n), k, largest=False) selects on CPU (unsorted SET order).
 * Valid for the nth_element branch (k*64 > n) and 1 <= k <= n.  Returns 0, or -1 on bad arguments. */
int r3d_oracle_select_smallest(const float* score, int64_t n, int64_t k, int64_t* out) {
    if (!score || !out || n <= 0 || k <= 0 || k > n || k * 64 <= n) return -1;
    elem_t* q = (elem_t*)malloc((size_t)n * sizeof(elem_t));
    if (!q) return -1;
    for (int64_t j = 0; j < n; ++j) { q[j].v = score[j]; q[j].i = j; }
    introselect(q, q + (k - 1), q + n, 2 * lg2(n));
    for (int64_t j = 0; j < k; ++j) out[j] = q[j].i;
    free(q);
    return 0;
}
