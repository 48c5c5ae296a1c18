/*
 * vb_oracle_decode.c -- CPU restatement of the token-passing searches (test
 * infrastructure only; see vb_oracle.h).  Reference paths relative to
 * /root/reference/kaldi-master/src/.
 *
 *   LatticeFasterDecoder  decoder/lattice-faster-decoder.{h,cc}
 *   FasterDecoder         decoder/faster-decoder.{h,cc}
 *   HashList iteration order  util/hash-list-inl.h:137-165
 *
 * The reference's result depends on the order in which it visits tokens (the online
 * tightening of next_cutoff, lattice-faster-decoder.cc:758-760) and that order is the
 * HashList order: buckets (state % hash_size) in order of first occupation, elements inside
 * a bucket in insertion order.  This file reproduces that order with its own data
 * structure (an insertion stamp per element and an occupation stamp per bucket, sorted when
 * the list is needed), so that cfg.exact_cutoff == 0 gives the reference's token sets in
 * every regime.  cfg.exact_cutoff == 1 is the order-independent variant (the GPU search's
 * admission = exact, and the 1-best search), == 3 the reference's admission rule in a canonical
 * order (the GPU lattice search's default, admission = ordered); see vb_oracle.h.
 */
#include "vb_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define VBO_INF INFINITY

void vbo_lat_cfg_default(vbo_lat_cfg *c) {
  /* lattice-faster-decoder.h:59-67 */
  c->beam = 16.0f; c->max_active = INT32_MAX; c->min_active = 200;
  c->lattice_beam = 10.0f; c->prune_interval = 25; c->beam_delta = 0.5f;
  c->hash_ratio = 2.0f; c->prune_scale = 0.1f; c->exact_cutoff = 0; c->adm_bucket_bits = 6;
}
void vbo_faster_cfg_default(vbo_faster_cfg *c) {
  /* faster-decoder.h:38-43 */
  c->beam = 16.0f; c->max_active = INT32_MAX; c->min_active = 20;
  c->beam_delta = 0.5f; c->hash_ratio = 2.0f; c->exact_cutoff = 0; c->adm_bucket_bits = 6;
}

/* ------------------------------------------------------------------------------------
 * Ordered state map: restates the observable behaviour of HashList<StateId, T>
 * (Find / Insert / GetList order / Clear / SetSize) without its pointer structure.
 * ---------------------------------------------------------------------------------- */
typedef struct {
  int32_t state, val;      /* key, payload (token index) */
  int64_t ins_seq;         /* insertion stamp */
  int32_t bucket;
} omap_elem;

typedef struct {
  size_t hash_size;        /* HashList::hash_size_ */
  /* per-bucket occupation stamp, valid when bucket_epoch == epoch */
  int64_t *bucket_occ; int32_t *bucket_epoch; size_t bucket_cap;
  int32_t epoch;
  int64_t seq;
  omap_elem *elems; int32_t n, cap;
  /* open addressing index: slot -> elem index + 1 (0 = empty), stamped by epoch */
  int32_t *slot_elem; int32_t *slot_epoch; size_t slots;  /* power of two */
} omap;

static void omap_init(omap *m, size_t hash_size) {
  memset(m, 0, sizeof(*m));
  m->hash_size = hash_size; m->epoch = 1;
  m->slots = 1 << 12;
  m->slot_elem = (int32_t *)calloc(m->slots, sizeof(int32_t));
  m->slot_epoch = (int32_t *)calloc(m->slots, sizeof(int32_t));
}
static void omap_free(omap *m) {
  free(m->bucket_occ); free(m->bucket_epoch); free(m->elems);
  free(m->slot_elem); free(m->slot_epoch);
}
static void omap_clear(omap *m) { m->epoch++; m->n = 0; }
static void omap_set_size(omap *m, size_t sz) { m->hash_size = sz; }
static size_t omap_slot(const omap *m, int32_t state) {
  return ((uint32_t)state * 2654435761u) & (m->slots - 1);
}
static void omap_rehash(omap *m) {
  size_t ns = m->slots * 2;
  free(m->slot_elem); free(m->slot_epoch);
  m->slot_elem = (int32_t *)calloc(ns, sizeof(int32_t));
  m->slot_epoch = (int32_t *)calloc(ns, sizeof(int32_t));
  m->slots = ns;
  for (int32_t i = 0; i < m->n; i++) {
    size_t s = omap_slot(m, m->elems[i].state);
    while (m->slot_epoch[s] == m->epoch) s = (s + 1) & (m->slots - 1);
    m->slot_epoch[s] = m->epoch; m->slot_elem[s] = i + 1;
  }
}
/* returns elem index or -1 */
static int32_t omap_find(const omap *m, int32_t state) {
  size_t s = omap_slot(m, state);
  while (m->slot_epoch[s] == m->epoch) {
    int32_t e = m->slot_elem[s] - 1;
    if (m->elems[e].state == state) return e;
    s = (s + 1) & (m->slots - 1);
  }
  return -1;
}
static void omap_insert(omap *m, int32_t state, int32_t val) {
  if ((size_t)(m->n + 1) * 2 > m->slots) omap_rehash(m);
  if (m->n == m->cap) {
    m->cap = m->cap ? m->cap * 2 : 1024;
    m->elems = (omap_elem *)realloc(m->elems, sizeof(omap_elem) * m->cap);
  }
  size_t b = (size_t)state % m->hash_size;
  if (b >= m->bucket_cap) {
    size_t nc = m->bucket_cap ? m->bucket_cap : 1024;
    while (nc <= b) nc *= 2;
    m->bucket_occ = (int64_t *)realloc(m->bucket_occ, sizeof(int64_t) * nc);
    m->bucket_epoch = (int32_t *)realloc(m->bucket_epoch, sizeof(int32_t) * nc);
    memset(m->bucket_epoch + m->bucket_cap, 0, sizeof(int32_t) * (nc - m->bucket_cap));
    m->bucket_cap = nc;
  }
  int64_t stamp = m->seq++;
  if (m->bucket_epoch[b] != m->epoch) { m->bucket_epoch[b] = m->epoch; m->bucket_occ[b] = stamp; }
  omap_elem *e = &m->elems[m->n];
  e->state = state; e->val = val; e->ins_seq = stamp; e->bucket = (int32_t)b;
  size_t s = omap_slot(m, state);
  while (m->slot_epoch[s] == m->epoch) s = (s + 1) & (m->slots - 1);
  m->slot_epoch[s] = m->epoch; m->slot_elem[s] = m->n + 1;
  m->n++;
}
typedef struct { int64_t k1, k2; int32_t state, val; } olist_item;
static int olist_cmp(const void *a, const void *b) {
  const olist_item *x = (const olist_item *)a, *y = (const olist_item *)b;
  if (x->k1 != y->k1) return x->k1 < y->k1 ? -1 : 1;
  if (x->k2 != y->k2) return x->k2 < y->k2 ? -1 : 1;
  return 0;
}
/* the list in HashList order; caller frees */
static olist_item *omap_list(const omap *m, int32_t *n_out) {
  olist_item *l = (olist_item *)malloc(sizeof(olist_item) * (m->n > 0 ? m->n : 1));
  for (int32_t i = 0; i < m->n; i++) {
    l[i].k1 = m->bucket_occ[m->elems[i].bucket]; l[i].k2 = m->elems[i].ins_seq;
    l[i].state = m->elems[i].state; l[i].val = m->elems[i].val;
  }
  qsort(l, m->n, sizeof(olist_item), olist_cmp);
  *n_out = m->n;
  return l;
}

static int cmp_f32(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}
static int cmp_f64(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

/* ------------------------------------------------------------------------------------
 * LatticeFasterDecoder
 * ---------------------------------------------------------------------------------- */
typedef struct {
  float tot_cost, extra_cost;
  int32_t links;   /* head of link list, -1 none */
  int32_t next;    /* next token of the same frame, -1 none */
  int32_t state;   /* HCLG state (the reference keeps it in the hash key only) */
} ltok;
typedef struct {
  int32_t next_tok, ilabel, olabel;
  float graph_cost, acoustic_cost;
  int32_t next;
} llink;
typedef struct { int32_t toks; int8_t must_prune_forward_links, must_prune_tokens; } lframe;

typedef struct {
  const vbo_fst *fst; const vbo_lat_cfg *cfg;
  const float *loglikes; int32_t T, P; const int32_t *tid2pdf; int32_t num_tids; float acwt;
  ltok *toks; int32_t ntoks, captoks, free_tok;
  llink *links; int32_t nlinks, caplinks, free_link;
  lframe *frames; int32_t nframes, capframes;   /* active_toks_ */
  float *cost_offsets; int32_t ncost;
  omap map;
  int32_t num_toks;            /* num_toks_ */
  /* final */
  int decoding_finalized; float final_relative_cost, final_best_cost;
  int final_costs_empty; float *final_cost_of_tok; /* indexed by token, inf when absent */
  int64_t n_tok_created, n_links_created;
  int32_t n_max_bound, n_min_bound;
  int32_t *queue; int32_t qn, qcap;
} ldec;

static int32_t new_tok(ldec *d, float tot, float extra, int32_t next, int32_t state) {
  int32_t i;
  if (d->free_tok >= 0) { i = d->free_tok; d->free_tok = d->toks[i].next; }
  else {
    if (d->ntoks == d->captoks) {
      d->captoks = d->captoks ? d->captoks * 2 : 4096;
      d->toks = (ltok *)realloc(d->toks, sizeof(ltok) * d->captoks);
    }
    i = d->ntoks++;
  }
  d->toks[i].tot_cost = tot; d->toks[i].extra_cost = extra; d->toks[i].links = -1;
  d->toks[i].next = next; d->toks[i].state = state;
  d->n_tok_created++;
  return i;
}
static void del_tok(ldec *d, int32_t i) { d->toks[i].next = d->free_tok; d->toks[i].state = -1; d->free_tok = i; }
static int32_t new_link(ldec *d, int32_t next_tok, int32_t il, int32_t ol, float g, float a,
                        int32_t next) {
  int32_t i;
  if (d->free_link >= 0) { i = d->free_link; d->free_link = d->links[i].next; }
  else {
    if (d->nlinks == d->caplinks) {
      d->caplinks = d->caplinks ? d->caplinks * 2 : 8192;
      d->links = (llink *)realloc(d->links, sizeof(llink) * d->caplinks);
    }
    i = d->nlinks++;
  }
  llink *l = &d->links[i];
  l->next_tok = next_tok; l->ilabel = il; l->olabel = ol; l->graph_cost = g;
  l->acoustic_cost = a; l->next = next;
  d->n_links_created++;
  return i;
}
static void del_link(ldec *d, int32_t i) { d->links[i].next = d->free_link; d->free_link = i; }
static void delete_forward_links(ldec *d, int32_t t) {
  int32_t l = d->toks[t].links;
  while (l >= 0) { int32_t m = d->links[l].next; del_link(d, l); l = m; }
  d->toks[t].links = -1;
}
static void frames_resize(ldec *d, int32_t n) {
  if (n > d->capframes) {
    d->capframes = d->capframes ? d->capframes * 2 : 256;
    if (d->capframes < n) d->capframes = n;
    d->frames = (lframe *)realloc(d->frames, sizeof(lframe) * d->capframes);
  }
  for (int32_t i = d->nframes; i < n; i++) {
    d->frames[i].toks = -1; d->frames[i].must_prune_forward_links = 1;
    d->frames[i].must_prune_tokens = 1;
  }
  d->nframes = n;
}

/* decodable-am-diag-gmm.h:142-145: scale * loglike(frame, TransitionIdToPdf(tid)) */
static inline float dec_loglike(const ldec *d, int32_t frame, int32_t tid) {
  return d->acwt * d->loglikes[(size_t)frame * d->P + d->tid2pdf[tid]];
}

/* lattice-faster-decoder.cc:259-295 */
static int32_t find_or_add_token(ldec *d, int32_t state, int32_t frame_plus_one, float tot_cost,
                                 int *changed) {
  int32_t e = omap_find(&d->map, state);
  if (e < 0) {
    int32_t t = new_tok(d, tot_cost, 0.0f, d->frames[frame_plus_one].toks, state);
    d->frames[frame_plus_one].toks = t;
    d->num_toks++;
    omap_insert(&d->map, state, t);
    if (changed) *changed = 1;
    return t;
  }
  int32_t t = d->map.elems[e].val;
  if (d->toks[t].tot_cost > tot_cost) { d->toks[t].tot_cost = tot_cost; if (changed) *changed = 1; }
  else if (changed) *changed = 0;
  return t;
}

/* lattice-faster-decoder.cc:796-858 */
static void process_nonemitting(ldec *d, float cutoff) {
  int32_t frame = d->nframes - 2;
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  d->qn = 0;
  if (n > d->qcap) { d->qcap = n * 2 + 16; d->queue = (int32_t *)realloc(d->queue, sizeof(int32_t) * d->qcap); }
  for (int32_t i = 0; i < n; i++) d->queue[d->qn++] = l[i].state;
  free(l);
  const vbo_fst *fst = d->fst;
  while (d->qn > 0) {
    int32_t state = d->queue[--d->qn];
    int32_t tok = d->map.elems[omap_find(&d->map, state)].val;
    float cur_cost = d->toks[tok].tot_cost;
    if (cur_cost > cutoff) continue;
    delete_forward_links(d, tok);
    for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
      const vbo_arc *arc = &fst->arcs[a];
      if (arc->ilabel == 0) {
        float graph_cost = arc->weight, tot_cost = cur_cost + graph_cost;
        if (tot_cost < cutoff) {
          int changed;
          int32_t nt = find_or_add_token(d, arc->nextstate, frame + 1, tot_cost, &changed);
          d->toks[tok].links = new_link(d, nt, 0, arc->olabel, graph_cost, 0.0f, d->toks[tok].links);
          if (changed) {
            if (d->qn == d->qcap) { d->qcap = d->qcap * 2 + 16; d->queue = (int32_t *)realloc(d->queue, sizeof(int32_t) * d->qcap); }
            d->queue[d->qn++] = arc->nextstate;
          }
        }
      }
    }
  }
}

/* lattice-faster-decoder.cc:618-685; list gives tot_costs in HashList order */
static float get_cutoff(ldec *d, const olist_item *l, int32_t n, float *adaptive_beam,
                        int32_t *best_idx) {
  const vbo_lat_cfg *c = d->cfg;
  float best = VBO_INF;
  *best_idx = -1;
  for (int32_t i = 0; i < n; i++) {
    float w = d->toks[l[i].val].tot_cost;
    if (w < best) { best = w; *best_idx = i; }
  }
  if (c->max_active == INT32_MAX && c->min_active == 0) { *adaptive_beam = c->beam; return best + c->beam; }
  float beam_cutoff = best + c->beam, min_active_cutoff = VBO_INF, max_active_cutoff = VBO_INF;
  float *tmp = (float *)malloc(sizeof(float) * (n > 0 ? n : 1));
  for (int32_t i = 0; i < n; i++) tmp[i] = d->toks[l[i].val].tot_cost;
  int sorted = 0;
  float ret;
  if ((size_t)n > (size_t)c->max_active) {
    qsort(tmp, n, sizeof(float), cmp_f32); sorted = 1;
    max_active_cutoff = tmp[c->max_active];   /* nth_element: element of rank max_active */
  }
  if (max_active_cutoff < beam_cutoff) {
    *adaptive_beam = max_active_cutoff - best + c->beam_delta;
    d->n_max_bound++;
    ret = max_active_cutoff;
  } else {
    if ((size_t)n > (size_t)c->min_active) {
      if (c->min_active == 0) min_active_cutoff = best;
      else {
        if (!sorted) qsort(tmp, n, sizeof(float), cmp_f32);
        min_active_cutoff = tmp[c->min_active];
      }
    }
    if (min_active_cutoff > beam_cutoff) {
      *adaptive_beam = min_active_cutoff - best + c->beam_delta;
      d->n_min_bound++;
      ret = min_active_cutoff;
    } else { *adaptive_beam = c->beam; ret = beam_cutoff; }
  }
  free(tmp);
  return ret;
}

/* lattice-faster-decoder.cc:687-777 */
static float process_emitting(ldec *d) {
  const vbo_lat_cfg *c = d->cfg; const vbo_fst *fst = d->fst;
  int32_t frame = d->nframes - 1;
  frames_resize(d, d->nframes + 1);
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  omap_clear(&d->map);
  float adaptive_beam; int32_t best_idx;
  float cur_cutoff = get_cutoff(d, l, n, &adaptive_beam, &best_idx);
  if (c->exact_cutoff && best_idx >= 0) {
    /* order-independent choice among equal-cost best tokens: the smallest state id (the
     * reference takes the first one in HashList order, lattice-faster-decoder.cc:636-641) */
    float bw = d->toks[l[best_idx].val].tot_cost;
    for (int32_t i = 0; i < n; i++)
      if (d->toks[l[i].val].tot_cost == bw && l[i].state < l[best_idx].state) best_idx = i;
  }
  /* PossiblyResizeHash :194-200 */
  size_t new_sz = (size_t)((float)n * c->hash_ratio);
  if (new_sz > d->map.hash_size) omap_set_size(&d->map, new_sz);

  float next_cutoff = VBO_INF, cost_offset = 0.0f;
  if (best_idx >= 0) {
    int32_t state = l[best_idx].state; float tot = d->toks[l[best_idx].val].tot_cost;
    cost_offset = -tot;
    for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
      const vbo_arc *arc = &fst->arcs[a];
      if (arc->ilabel != 0) {
        float new_weight = arc->weight + cost_offset - dec_loglike(d, frame, arc->ilabel) + tot;
        if (new_weight + adaptive_beam < next_cutoff) next_cutoff = new_weight + adaptive_beam;
      }
    }
  }
  if (frame + 1 > d->ncost) { d->cost_offsets = (float *)realloc(d->cost_offsets, sizeof(float) * (frame + 1)); d->ncost = frame + 1; }
  d->cost_offsets[frame] = cost_offset;

  /* exact_cutoff == 2: the admission bound is the reference's INITIAL next_cutoff (from the best token's arcs, :712-724) and stays
   * there: every arc the reference can admit in any visiting order is admitted (its running bound only falls from this value);
   * the value handed to ProcessNonemitting is the final one, as in the other modes */
  const float admit_cutoff = next_cutoff;
  /* exact_cutoff == 3: per-bucket minima of tot + adaptive_beam, then the exclusive prefix minimum over the buckets (starting
   * from the initial next_cutoff): pre[b] is the bound the arcs leaving a token of bucket b are admitted against */
  float *pre = NULL;
  int32_t bbits = c->adm_bucket_bits < 1 ? 1 : (c->adm_bucket_bits > 20 ? 20 : c->adm_bucket_bits);
  if (c->exact_cutoff == 3) {
    pre = (float *)malloc(sizeof(float) * ((size_t)1 << bbits));
    for (size_t b = 0; b < ((size_t)1 << bbits); b++) pre[b] = VBO_INF;
  }
#define VBO_ADM_BUCKET(state) ((uint32_t)((uint32_t)(state) * 2654435761u) >> (32 - bbits))
  if (c->exact_cutoff) {
    /* order-independent variants: next_cutoff takes its final value before any admission */
    for (int32_t i = 0; i < n; i++) {
      int32_t tok = l[i].val, state = l[i].state;
      if (d->toks[tok].tot_cost <= cur_cutoff)
        for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
          const vbo_arc *arc = &fst->arcs[a];
          if (arc->ilabel != 0) {
            float ac_cost = cost_offset - dec_loglike(d, frame, arc->ilabel),
                  tot_cost = d->toks[tok].tot_cost + ac_cost + arc->weight;
            if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
            if (pre && tot_cost + adaptive_beam < pre[VBO_ADM_BUCKET(state)]) pre[VBO_ADM_BUCKET(state)] = tot_cost + adaptive_beam;
          }
        }
    }
  }
  if (pre) {
    float run = admit_cutoff;
    for (size_t b = 0; b < ((size_t)1 << bbits); b++) { const float m = pre[b]; pre[b] = run; if (m < run) run = m; }
  }
  for (int32_t i = 0; i < n; i++) {
    int32_t tok = l[i].val, state = l[i].state;
    if (d->toks[tok].tot_cost <= cur_cutoff) {
      for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
        const vbo_arc *arc = &fst->arcs[a];
        if (arc->ilabel != 0) {
          float ac_cost = cost_offset - dec_loglike(d, frame, arc->ilabel),
                graph_cost = arc->weight, cur_cost = d->toks[tok].tot_cost,
                tot_cost = cur_cost + ac_cost + graph_cost;
          if (c->exact_cutoff == 3) { if (tot_cost > pre[VBO_ADM_BUCKET(state)]) continue; }
          else if (c->exact_cutoff == 2) { if (tot_cost > admit_cutoff) continue; }
          else if (tot_cost > next_cutoff) continue;
          else if (tot_cost + adaptive_beam < next_cutoff) next_cutoff = tot_cost + adaptive_beam;
          int32_t nt = find_or_add_token(d, arc->nextstate, frame + 1, tot_cost, NULL);
          d->toks[tok].links = new_link(d, nt, arc->ilabel, arc->olabel, graph_cost, ac_cost, d->toks[tok].links);
        }
      }
    }
  }
#undef VBO_ADM_BUCKET
  free(pre);
  free(l);
  return next_cutoff;
}

/* lattice-faster-decoder.cc:300-371 */
static void prune_forward_links(ldec *d, int32_t f, int *extra_costs_changed, int *links_pruned,
                                float delta) {
  *extra_costs_changed = 0; *links_pruned = 0;
  float lattice_beam = d->cfg->lattice_beam;
  int changed = 1;
  while (changed) {
    changed = 0;
    for (int32_t t = d->frames[f].toks; t >= 0; t = d->toks[t].next) {
      ltok *tok = &d->toks[t];
      int32_t link = tok->links, prev = -1;
      float tok_extra = VBO_INF;
      while (link >= 0) {
        llink *L = &d->links[link];
        const ltok *nt = &d->toks[L->next_tok];
        float link_extra = nt->extra_cost + ((tok->tot_cost + L->acoustic_cost + L->graph_cost) - nt->tot_cost);
        if (link_extra > lattice_beam) {
          int32_t nl = L->next;
          if (prev >= 0) d->links[prev].next = nl; else tok->links = nl;
          del_link(d, link);
          link = nl;
          *links_pruned = 1;
        } else {
          if (link_extra < 0.0f) link_extra = 0.0f;
          if (link_extra < tok_extra) tok_extra = link_extra;
          prev = link; link = L->next;
        }
      }
      if (fabsf(tok_extra - tok->extra_cost) > delta) changed = 1;
      tok->extra_cost = tok_extra;
    }
    if (changed) *extra_costs_changed = 1;
  }
}

/* lattice-faster-decoder.cc:477-496 */
static void prune_tokens_for_frame(ldec *d, int32_t f) {
  int32_t tok = d->frames[f].toks, prev = -1;
  while (tok >= 0) {
    int32_t next = d->toks[tok].next;
    if (d->toks[tok].extra_cost == VBO_INF) {
      if (prev >= 0) d->toks[prev].next = next; else d->frames[f].toks = next;
      delete_forward_links(d, tok);  /* none survive (extra == inf), keeps the pool tidy */
      del_tok(d, tok);
      d->num_toks--;
    } else prev = tok;
    tok = next;
  }
}

/* lattice-faster-decoder.cc:503-530 */
static void prune_active_tokens(ldec *d, float delta) {
  int32_t cur = d->nframes - 1;
  for (int32_t f = cur - 1; f >= 0; f--) {
    if (d->frames[f].must_prune_forward_links) {
      int ecc = 0, lp = 0;
      prune_forward_links(d, f, &ecc, &lp, delta);
      if (ecc && f > 0) d->frames[f - 1].must_prune_forward_links = 1;
      if (lp) d->frames[f].must_prune_tokens = 1;
      d->frames[f].must_prune_forward_links = 0;
    }
    if (f + 1 < cur && d->frames[f + 1].must_prune_tokens) {
      prune_tokens_for_frame(d, f + 1);
      d->frames[f + 1].must_prune_tokens = 0;
    }
  }
}

static int approx_equal(float a, float b, float tol) {
  /* base/kaldi-math.h:262-270 */
  if (a == b) return 1;
  float diff = fabsf(a - b);
  if (diff == VBO_INF || diff != diff) return 0;
  return diff <= tol * (fabsf(a) + fabsf(b));
}

/* lattice-faster-decoder.cc:532-572 + 376-458 */
static void prune_forward_links_final(ldec *d) {
  int32_t f = d->nframes - 1;
  float lattice_beam = d->cfg->lattice_beam;
  /* ComputeFinalCosts over the tokens of the last frame (the current hash) */
  float best_cost = VBO_INF, best_with_final = VBO_INF;
  d->final_cost_of_tok = (float *)malloc(sizeof(float) * (d->ntoks > 0 ? d->ntoks : 1));
  for (int32_t i = 0; i < d->ntoks; i++) d->final_cost_of_tok[i] = VBO_INF;
  d->final_costs_empty = 1;
  for (int32_t i = 0; i < d->map.n; i++) {
    int32_t state = d->map.elems[i].state, tok = d->map.elems[i].val;
    float final_cost = d->fst->final[state];
    float cost = d->toks[tok].tot_cost, cwf = cost + final_cost;
    if (cost < best_cost) best_cost = cost;
    if (cwf < best_with_final) best_with_final = cwf;
    if (final_cost != VBO_INF) { d->final_cost_of_tok[tok] = final_cost; d->final_costs_empty = 0; }
  }
  if (best_cost == VBO_INF && best_with_final == VBO_INF) d->final_relative_cost = VBO_INF;
  else d->final_relative_cost = best_with_final - best_cost;
  d->final_best_cost = (best_with_final != VBO_INF) ? best_with_final : best_cost;
  d->decoding_finalized = 1;
  omap_clear(&d->map);

  int changed = 1; float delta = 1.0e-05f;
  while (changed) {
    changed = 0;
    for (int32_t t = d->frames[f].toks; t >= 0; t = d->toks[t].next) {
      ltok *tok = &d->toks[t];
      float final_cost;
      if (d->final_costs_empty) final_cost = 0.0f;
      else final_cost = d->final_cost_of_tok[t];
      float tok_extra = tok->tot_cost + final_cost - d->final_best_cost;
      int32_t link = tok->links, prev = -1;
      while (link >= 0) {
        llink *L = &d->links[link];
        const ltok *nt = &d->toks[L->next_tok];
        float link_extra = nt->extra_cost + ((tok->tot_cost + L->acoustic_cost + L->graph_cost) - nt->tot_cost);
        if (link_extra > lattice_beam) {
          int32_t nl = L->next;
          if (prev >= 0) d->links[prev].next = nl; else tok->links = nl;
          del_link(d, link);
          link = nl;
        } else {
          if (link_extra < 0.0f) link_extra = 0.0f;
          if (link_extra < tok_extra) tok_extra = link_extra;
          prev = link; link = L->next;
        }
      }
      if (tok_extra > lattice_beam) tok_extra = VBO_INF;
      if (!approx_equal(tok->extra_cost, tok_extra, delta)) changed = 1;
      tok->extra_cost = tok_extra;
    }
  }
}

/* lattice-faster-decoder.cc:600-615 */
static void finalize_decoding(ldec *d) {
  int32_t final_fpo = d->nframes - 1;
  prune_forward_links_final(d);
  for (int32_t f = final_fpo - 1; f >= 0; f--) {
    int b1, b2;
    prune_forward_links(d, f, &b1, &b2, 0.0f);
    prune_tokens_for_frame(d, f + 1);
  }
  prune_tokens_for_frame(d, 0);
}

/* ---- result extraction ---- */
typedef struct { int32_t frame, state, tok; } tok_key;
static int tok_key_cmp(const void *a, const void *b) {
  const tok_key *x = (const tok_key *)a, *y = (const tok_key *)b;
  if (x->frame != y->frame) return x->frame < y->frame ? -1 : 1;
  if (x->state != y->state) return x->state < y->state ? -1 : 1;
  return 0;
}
typedef struct { int32_t src, dst, il, ol; float g, a; } link_rec;
static int link_rec_cmp(const void *a, const void *b) {
  const link_rec *x = (const link_rec *)a, *y = (const link_rec *)b;
  if (x->src != y->src) return x->src < y->src ? -1 : 1;
  if (x->dst != y->dst) return x->dst < y->dst ? -1 : 1;
  if (x->il != y->il) return x->il < y->il ? -1 : 1;
  if (x->ol != y->ol) return x->ol < y->ol ? -1 : 1;
  if (x->g != y->g) return x->g < y->g ? -1 : 1;
  if (x->a != y->a) return x->a < y->a ? -1 : 1;
  return 0;
}

/* LatticeWeight natural order (fstext/lattice-weight.h Compare): smaller sum wins, ties
 * broken on the first component.  returns 1 when (a1,a2) is strictly better. */
static int lw_better(float a1, float a2, float b1, float b2) {
  float fa = a1 + a2, fb = b1 + b2;
  if (fa < fb) return 1;
  if (fa > fb) return 0;
  return a1 < b1;
}

/* GetRawLattice (:110-192) restricted to what the tests compare + GetBestPath (:100-106)
 * + GetLinearSymbolSequence (fstext/fstext-utils-inl.h:178-216). */
static void extract_result(ldec *d, vbo_lat_result *r) {
  int32_t nf = d->nframes;   /* frames_plus_one count */
  int32_t n = 0;
  for (int32_t f = 0; f < nf; f++) for (int32_t t = d->frames[f].toks; t >= 0; t = d->toks[t].next) n++;
  tok_key *keys = (tok_key *)malloc(sizeof(tok_key) * (n > 0 ? n : 1));
  int32_t k = 0;
  for (int32_t f = 0; f < nf; f++)
    for (int32_t t = d->frames[f].toks; t >= 0; t = d->toks[t].next) {
      keys[k].frame = f; keys[k].state = d->toks[t].state; keys[k].tok = t; k++;
    }
  qsort(keys, n, sizeof(tok_key), tok_key_cmp);
  int32_t *idx_of_tok = (int32_t *)malloc(sizeof(int32_t) * (d->ntoks > 0 ? d->ntoks : 1));
  for (int32_t i = 0; i < d->ntoks; i++) idx_of_tok[i] = -1;
  r->num_toks = n;
  r->tok_frame = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
  r->tok_state = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
  r->tok_cost = (float *)malloc(sizeof(float) * (n + 1));
  r->tok_extra = (float *)malloc(sizeof(float) * (n + 1));
  r->tok_final = (float *)malloc(sizeof(float) * (n + 1));
  int64_t nl = 0;
  for (int32_t i = 0; i < n; i++) {
    int32_t t = keys[i].tok;
    idx_of_tok[t] = i;
    r->tok_frame[i] = keys[i].frame; r->tok_state[i] = keys[i].state;
    r->tok_cost[i] = d->toks[t].tot_cost; r->tok_extra[i] = d->toks[t].extra_cost;
    /* final weight as GetRawLattice sets it (:176-185) */
    float fin = VBO_INF;
    if (keys[i].frame == nf - 1) {
      if (!d->final_costs_empty) fin = d->final_cost_of_tok[t];
      else fin = 0.0f;
    }
    r->tok_final[i] = fin;
    for (int32_t l = d->toks[t].links; l >= 0; l = d->links[l].next) nl++;
  }
  link_rec *recs = (link_rec *)malloc(sizeof(link_rec) * (nl > 0 ? nl : 1));
  int64_t j = 0;
  for (int32_t i = 0; i < n; i++) {
    int32_t t = keys[i].tok, f = keys[i].frame;
    for (int32_t l = d->toks[t].links; l >= 0; l = d->links[l].next) {
      const llink *L = &d->links[l];
      float cost_offset = 0.0f;
      if (L->ilabel != 0) cost_offset = d->cost_offsets[f];
      recs[j].src = i; recs[j].dst = idx_of_tok[L->next_tok]; recs[j].il = L->ilabel;
      recs[j].ol = L->olabel; recs[j].g = L->graph_cost; recs[j].a = L->acoustic_cost - cost_offset;
      j++;
    }
  }
  qsort(recs, nl, sizeof(link_rec), link_rec_cmp);
  r->num_links = (int32_t)nl;
  r->link_src = (int32_t *)malloc(sizeof(int32_t) * (nl + 1)); r->link_dst = (int32_t *)malloc(sizeof(int32_t) * (nl + 1));
  r->link_ilabel = (int32_t *)malloc(sizeof(int32_t) * (nl + 1)); r->link_olabel = (int32_t *)malloc(sizeof(int32_t) * (nl + 1));
  r->link_graph = (float *)malloc(sizeof(float) * (nl + 1)); r->link_acoustic = (float *)malloc(sizeof(float) * (nl + 1));
  for (int64_t i = 0; i < nl; i++) {
    r->link_src[i] = recs[i].src; r->link_dst[i] = recs[i].dst; r->link_ilabel[i] = recs[i].il;
    r->link_olabel[i] = recs[i].ol; r->link_graph[i] = recs[i].g; r->link_acoustic[i] = recs[i].a;
  }

  /* single shortest path over the raw lattice (a DAG: links go to a later frame, or within
   * a frame along epsilon arcs).  Distances are LatticeWeight pairs accumulated from the
   * start; relax to a fixed point frame by frame. */
  float *d1 = (float *)malloc(sizeof(float) * (n + 1)), *d2 = (float *)malloc(sizeof(float) * (n + 1));
  int64_t *par = (int64_t *)malloc(sizeof(int64_t) * (n + 1));
  for (int32_t i = 0; i < n; i++) { d1[i] = VBO_INF; d2[i] = VBO_INF; par[i] = -1; }
  r->num_ali = 0; r->num_words = 0; r->ali = NULL; r->words = NULL;
  r->best_graph = VBO_INF; r->best_acoustic = VBO_INF;
  /* start token: the frame-0 token holding the start state (first token created) */
  int32_t start = -1;
  for (int32_t i = 0; i < n && r->tok_frame[i] == 0; i++) if (r->tok_state[i] == d->fst->start) start = i;
  if (start >= 0) {
    d1[start] = 0.0f; d2[start] = 0.0f;
    /* links sorted by src => by frame; iterate each frame's link range to a fixed point */
    int64_t lo = 0;
    while (lo < nl) {
      int32_t f = r->tok_frame[recs[lo].src];
      int64_t hi = lo;
      while (hi < nl && r->tok_frame[recs[hi].src] == f) hi++;
      int again = 1;
      while (again) {
        again = 0;
        for (int64_t i = lo; i < hi; i++) {
          int32_t s = recs[i].src, t = recs[i].dst;
          if (d1[s] == VBO_INF && d2[s] == VBO_INF) continue;
          float n1 = d1[s] + recs[i].g, n2 = d2[s] + recs[i].a;
          if (lw_better(n1, n2, d1[t], d2[t])) {
            d1[t] = n1; d2[t] = n2; par[t] = i;
            if (r->tok_frame[t] == f) again = 1;
          }
        }
      }
      lo = hi;
    }
    int32_t best = -1; float b1 = VBO_INF, b2 = VBO_INF;
    for (int32_t i = 0; i < n; i++) {
      if (r->tok_frame[i] != nf - 1 || r->tok_final[i] == VBO_INF) continue;
      if (d1[i] == VBO_INF && d2[i] == VBO_INF) continue;
      float f1 = d1[i] + r->tok_final[i], f2 = d2[i] + 0.0f;
      if (best < 0 || lw_better(f1, f2, b1, b2)) { best = i; b1 = f1; b2 = f2; }
    }
    if (best >= 0) {
      int32_t len = 0;
      for (int32_t t = best; par[t] >= 0; t = recs[par[t]].src) len++;
      int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (len + 1));
      int32_t p = len;
      for (int32_t t = best; par[t] >= 0; t = recs[par[t]].src) path[--p] = par[t];
      r->ali = (int32_t *)malloc(sizeof(int32_t) * (len + 1));
      r->words = (int32_t *)malloc(sizeof(int32_t) * (len + 1));
      float t1 = 0.0f, t2 = 0.0f;
      for (int32_t i = 0; i < len; i++) {
        const link_rec *L = &recs[path[i]];
        t1 = L->g + t1; t2 = L->a + t2;
        if (L->il != 0) r->ali[r->num_ali++] = L->il;
        if (L->ol != 0) r->words[r->num_words++] = L->ol;
      }
      t1 = r->tok_final[best] + t1; t2 = 0.0f + t2;
      r->best_graph = t1; r->best_acoustic = t2;
      free(path);
    }
  }
  free(d1); free(d2); free(par); free(recs); free(keys); free(idx_of_tok);
}

int vbo_decode_lattice_faster(const vbo_fst *fst, const vbo_lat_cfg *cfg, const float *loglikes,
                              int32_t T, int32_t P, const int32_t *tid2pdf, int32_t num_tids,
                              float acwt, vbo_lat_result *res) {
  memset(res, 0, sizeof(*res));
  if (fst->start < 0 || fst->start >= fst->num_states) return -1;
  ldec dd; ldec *d = &dd;
  memset(d, 0, sizeof(*d));
  d->fst = fst; d->cfg = cfg; d->loglikes = loglikes; d->T = T; d->P = P;
  d->tid2pdf = tid2pdf; d->num_tids = num_tids; d->acwt = acwt;
  d->free_tok = -1; d->free_link = -1;
  omap_init(&d->map, 1000);  /* toks_.SetSize(1000), :38 */
  /* InitDecoding :56-73 */
  frames_resize(d, 1);
  int32_t st = new_tok(d, 0.0f, 0.0f, -1, fst->start);
  d->frames[0].toks = st;
  omap_insert(&d->map, fst->start, st);
  d->num_toks++;
  process_nonemitting(d, cfg->beam);
  /* Decode :78-96; IsLastFrame(f) == (f == T-1) */
  while ((d->nframes - 1) - 1 != T - 1) {
    if ((d->nframes - 1) % cfg->prune_interval == 0)
      prune_active_tokens(d, cfg->lattice_beam * cfg->prune_scale);
    float cutoff = process_emitting(d);
    process_nonemitting(d, cutoff);
  }
  finalize_decoding(d);
  res->ok = (d->nframes > 0 && d->frames[d->nframes - 1].toks >= 0);
  res->num_frames = d->nframes - 1;
  res->final_relative_cost = d->final_relative_cost;
  res->final_best_cost = d->final_best_cost;
  res->reached_final = d->final_relative_cost != VBO_INF;
  res->n_tok_created = d->n_tok_created; res->n_links_created = d->n_links_created;
  res->n_frames_maxactive_bound = d->n_max_bound; res->n_frames_minactive_bound = d->n_min_bound;
  res->cost_offsets = (float *)malloc(sizeof(float) * (d->ncost > 0 ? d->ncost : 1));
  memcpy(res->cost_offsets, d->cost_offsets, sizeof(float) * d->ncost);
  extract_result(d, res);
  free(d->toks); free(d->links); free(d->frames); free(d->cost_offsets);
  free(d->final_cost_of_tok); free(d->queue);
  omap_free(&d->map);
  return 0;
}

void vbo_lat_result_free(vbo_lat_result *r) {
  free(r->tok_frame); free(r->tok_state); free(r->tok_cost); free(r->tok_extra); free(r->tok_final);
  free(r->link_src); free(r->link_dst); free(r->link_ilabel); free(r->link_olabel);
  free(r->link_graph); free(r->link_acoustic); free(r->ali); free(r->words); free(r->cost_offsets);
  memset(r, 0, sizeof(*r));
}

/* ------------------------------------------------------------------------------------
 * FasterDecoder (faster-decoder.cc).  Tokens carry a back pointer and a double cost.
 * Tokens are never freed here (the reference's reference counting only bounds memory).
 * ---------------------------------------------------------------------------------- */
typedef struct { int32_t ilabel, olabel, nextstate; float weight; int32_t prev; double cost; } ftok;
typedef struct {
  const vbo_fst *fst; const vbo_faster_cfg *cfg;
  const float *loglikes; int32_t P; const int32_t *tid2pdf; float acwt;
  ftok *toks; int32_t ntoks, captoks;
  omap map;
  int32_t *queue; int32_t qn, qcap;
} fdec;

static int32_t f_new_tok(fdec *d, const vbo_arc *arc, float ac_cost, int has_ac, int32_t prev) {
  if (d->ntoks == d->captoks) {
    d->captoks = d->captoks ? d->captoks * 2 : 4096;
    d->toks = (ftok *)realloc(d->toks, sizeof(ftok) * d->captoks);
  }
  ftok *t = &d->toks[d->ntoks];
  t->ilabel = arc->ilabel; t->olabel = arc->olabel; t->nextstate = arc->nextstate;
  t->weight = arc->weight; t->prev = prev;
  /* faster-decoder.h:116-133 */
  if (has_ac) t->cost = (prev >= 0 ? d->toks[prev].cost + arc->weight + ac_cost : (double)arc->weight + ac_cost);
  else t->cost = (prev >= 0 ? d->toks[prev].cost + arc->weight : (double)arc->weight);
  return d->ntoks++;
}

/* faster-decoder.cc:306-346 */
static void f_process_nonemitting(fdec *d, double cutoff) {
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  d->qn = 0;
  if (n > d->qcap) { d->qcap = 2 * n + 16; d->queue = (int32_t *)realloc(d->queue, sizeof(int32_t) * d->qcap); }
  for (int32_t i = 0; i < n; i++) d->queue[d->qn++] = l[i].state;
  free(l);
  const vbo_fst *fst = d->fst;
  while (d->qn > 0) {
    int32_t state = d->queue[--d->qn];
    int32_t e = omap_find(&d->map, state);
    int32_t tok = d->map.elems[e].val;
    if (d->toks[tok].cost > cutoff) continue;
    for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
      const vbo_arc *arc = &fst->arcs[a];
      if (arc->ilabel != 0) continue;
      int32_t nt = f_new_tok(d, arc, 0.0f, 0, tok);
      if (d->toks[nt].cost > cutoff) { d->ntoks--; continue; }
      int32_t ef = omap_find(&d->map, arc->nextstate);
      int push = 0;
      if (ef < 0) { omap_insert(&d->map, arc->nextstate, nt); push = 1; }
      else if (d->toks[d->map.elems[ef].val].cost > d->toks[nt].cost) { d->map.elems[ef].val = nt; push = 1; }
      if (push) {
        if (d->qn == d->qcap) { d->qcap = 2 * d->qcap + 16; d->queue = (int32_t *)realloc(d->queue, sizeof(int32_t) * d->qcap); }
        d->queue[d->qn++] = arc->nextstate;
      }
    }
  }
}

/* faster-decoder.cc:151-213 */
static double f_get_cutoff(fdec *d, const olist_item *l, int32_t n, float *adaptive_beam, int32_t *best_idx) {
  const vbo_faster_cfg *c = d->cfg;
  double best = INFINITY;
  *best_idx = -1;
  for (int32_t i = 0; i < n; i++) {
    double w = d->toks[l[i].val].cost;
    if (w < best) { best = w; *best_idx = i; }
  }
  if (c->max_active == INT32_MAX && c->min_active == 0) { *adaptive_beam = c->beam; return best + c->beam; }
  double beam_cutoff = best + c->beam, min_active_cutoff = INFINITY, max_active_cutoff = INFINITY;
  /* faster-decoder.h:174: tmp_array_ is a std::vector<BaseFloat> -- the costs, doubles in the tokens, are NARROWED TO FLOAT when GetCutoff
   * collects them (:171), so both order statistics are float32-rounded costs (promoted back to double, :183, :196), while best_cost and
   * beam_cutoff stay exact doubles.  Where a cutoff falls among (nearly) equal costs -- the fan-out of a word end over the phone-start
   * states by weight-0 epsilon arcs leaves dozens of exactly equal costs, and min-active = 20 puts the cutoff right there -- the rounding
   * decides whether `cost < cutoff` (:262) expands all of them or none: found in round 5 by the 512-utterance gmm-decode-faster fixture
   * (tests/golden/speech_faster512.npz), where a search with exact-double statistics loses the path on most utterances. */
  double *tmp = (double *)malloc(sizeof(double) * (n > 0 ? n : 1));
  for (int32_t i = 0; i < n; i++) tmp[i] = (double)(float)d->toks[l[i].val].cost;
  int sorted = 0; double ret;
  if ((size_t)n > (size_t)c->max_active) { qsort(tmp, n, sizeof(double), cmp_f64); sorted = 1; max_active_cutoff = tmp[c->max_active]; }
  if (max_active_cutoff < beam_cutoff) {
    *adaptive_beam = (float)(max_active_cutoff - best + c->beam_delta);
    ret = max_active_cutoff;
  } else {
    if ((size_t)n > (size_t)c->min_active) {
      if (c->min_active == 0) min_active_cutoff = best;
      else { if (!sorted) qsort(tmp, n, sizeof(double), cmp_f64); min_active_cutoff = tmp[c->min_active]; }
    }
    if (min_active_cutoff > beam_cutoff) {
      *adaptive_beam = (float)(min_active_cutoff - best + c->beam_delta);
      ret = min_active_cutoff;
    } else { *adaptive_beam = c->beam; ret = beam_cutoff; }
  }
  free(tmp);
  return ret;
}

/* faster-decoder.cc:224-303 */
static double f_process_emitting(fdec *d, int32_t frame) {
  const vbo_fst *fst = d->fst;
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  omap_clear(&d->map);
  float adaptive_beam; int32_t best_idx;
  double weight_cutoff = f_get_cutoff(d, l, n, &adaptive_beam, &best_idx);
  size_t new_sz = (size_t)((float)n * d->cfg->hash_ratio);
  if (new_sz > d->map.hash_size) omap_set_size(&d->map, new_sz);
  double next_cutoff = INFINITY;
  if (best_idx >= 0) {
    int32_t state = l[best_idx].state; int32_t tok = l[best_idx].val;
    for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
      const vbo_arc *arc = &fst->arcs[a];
      if (arc->ilabel != 0) {
        float ac_cost = -(d->acwt * d->loglikes[(size_t)frame * d->P + d->tid2pdf[arc->ilabel]]);
        double new_weight = arc->weight + d->toks[tok].cost + ac_cost;
        if (new_weight + adaptive_beam < next_cutoff) next_cutoff = new_weight + adaptive_beam;
      }
    }
  }
  for (int32_t i = 0; i < n; i++) {
    int32_t state = l[i].state, tok = l[i].val;
    if (d->toks[tok].cost < weight_cutoff) {
      for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
        const vbo_arc *arc = &fst->arcs[a];
        if (arc->ilabel == 0) continue;
        float ac_cost = -(d->acwt * d->loglikes[(size_t)frame * d->P + d->tid2pdf[arc->ilabel]]);
        double new_weight = arc->weight + d->toks[tok].cost + ac_cost;
        if (new_weight < next_cutoff) {
          int32_t nt = f_new_tok(d, arc, ac_cost, 1, tok);
          int32_t ef = omap_find(&d->map, arc->nextstate);
          if (new_weight + adaptive_beam < next_cutoff) next_cutoff = new_weight + adaptive_beam;
          if (ef < 0) omap_insert(&d->map, arc->nextstate, nt);
          else if (d->toks[d->map.elems[ef].val].cost > d->toks[nt].cost) d->map.elems[ef].val = nt;
        }
      }
    }
  }
  free(l);
  return next_cutoff;
}

/* ---- order-independent mode (cfg->exact_cutoff): one frame = entries keyed by state, each
 * holding the cheapest way in {cost, arc index, where it comes from}; tokens are made from the
 * entries once the epsilon closure has reached its fixed point. */
typedef struct { int32_t state; double cost; int64_t arc; int32_t prev_tok; int32_t prev_state; int32_t tok; } fentry;
typedef struct { fentry *e; int32_t n, cap; int32_t *slot; int32_t nslot; } fentries;   /* slot: state -> entry, open addressing */
static void fe_init(fentries *f) { memset(f, 0, sizeof(*f)); f->nslot = 1024; f->slot = (int32_t *)malloc(sizeof(int32_t) * f->nslot); memset(f->slot, 0xFF, sizeof(int32_t) * f->nslot); }
static void fe_free(fentries *f) { free(f->e); free(f->slot); }
static void fe_clear(fentries *f) { f->n = 0; memset(f->slot, 0xFF, sizeof(int32_t) * f->nslot); }
static int32_t fe_find(const fentries *f, int32_t state) {
  uint32_t h = ((uint32_t)state * 2654435761u) & (uint32_t)(f->nslot - 1);
  while (f->slot[h] >= 0) { if (f->e[f->slot[h]].state == state) return f->slot[h]; h = (h + 1) & (uint32_t)(f->nslot - 1); }
  return -1;
}
static void fe_put_slot(fentries *f, int32_t idx) {
  uint32_t h = ((uint32_t)f->e[idx].state * 2654435761u) & (uint32_t)(f->nslot - 1);
  while (f->slot[h] >= 0) h = (h + 1) & (uint32_t)(f->nslot - 1);
  f->slot[h] = idx;
}
/* returns 1 if the entry of `state` is new or got a lower cost (its epsilon arcs must be (re-)expanded) */
static int fe_relax(fentries *f, int32_t state, double cost, int64_t arc, int32_t prev_tok, int32_t prev_state) {
  int32_t i = fe_find(f, state);
  if (i < 0) {
    if (f->n == f->cap) { f->cap = f->cap ? 2 * f->cap : 1024; f->e = (fentry *)realloc(f->e, sizeof(fentry) * f->cap); }
    if (2 * (f->n + 1) > f->nslot) {
      f->nslot *= 2; f->slot = (int32_t *)realloc(f->slot, sizeof(int32_t) * f->nslot); memset(f->slot, 0xFF, sizeof(int32_t) * f->nslot);
      for (int32_t k = 0; k < f->n; k++) fe_put_slot(f, k);
    }
    fentry *e = &f->e[f->n];
    e->state = state; e->cost = cost; e->arc = arc; e->prev_tok = prev_tok; e->prev_state = prev_state; e->tok = -1;
    fe_put_slot(f, f->n++);
    return 1;
  }
  fentry *e = &f->e[i];
  if (cost < e->cost || (cost == e->cost && arc < e->arc)) {
    const int lower = cost < e->cost;
    e->cost = cost; e->arc = arc; e->prev_tok = prev_tok; e->prev_state = prev_state;
    return lower;
  }
  return 0;
}
/* closure over input-epsilon arcs to the fixed point, then one token per entry */
static void fx_close_and_make(fdec *d, fentries *f, double cutoff) {
  const vbo_fst *fst = d->fst;
  int32_t qn = 0, qcap = f->n + 16;
  int32_t *q = (int32_t *)malloc(sizeof(int32_t) * qcap);
  for (int32_t i = 0; i < f->n; i++) q[qn++] = f->e[i].state;
  while (qn > 0) {
    const int32_t state = q[--qn];
    const double c = f->e[fe_find(f, state)].cost;
    if (c > cutoff) continue;
    for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
      const vbo_arc *arc = &fst->arcs[a];
      if (arc->ilabel != 0) continue;
      const double nc = c + arc->weight;
      if (nc > cutoff) continue;
      if (fe_relax(f, arc->nextstate, nc, a, -1, state)) {
        if (qn == qcap) { qcap = 2 * qcap + 16; q = (int32_t *)realloc(q, sizeof(int32_t) * qcap); }
        q[qn++] = arc->nextstate;
      }
    }
  }
  free(q);
  /* tokens: indices first (a token's predecessor may be any entry of the same frame), then contents */
  for (int32_t i = 0; i < f->n; i++) {
    if (d->ntoks == d->captoks) { d->captoks = d->captoks ? d->captoks * 2 : 4096; d->toks = (ftok *)realloc(d->toks, sizeof(ftok) * d->captoks); }
    f->e[i].tok = d->ntoks++;
  }
  omap_clear(&d->map);
  for (int32_t i = 0; i < f->n; i++) {
    const fentry *e = &f->e[i];
    ftok *t = &d->toks[e->tok];
    if (e->arc >= 0) { const vbo_arc *arc = &fst->arcs[e->arc]; t->ilabel = arc->ilabel; t->olabel = arc->olabel; t->weight = arc->weight; }
    else { t->ilabel = 0; t->olabel = 0; t->weight = 0.0f; }
    t->nextstate = e->state; t->cost = e->cost;
    t->prev = e->arc < 0 ? -1 : (e->prev_tok >= 0 ? e->prev_tok : f->e[fe_find(f, e->prev_state)].tok);
    omap_insert(&d->map, e->state, e->tok);
  }
}
static double fx_process_emitting(fdec *d, fentries *f, int32_t frame) {
  const vbo_fst *fst = d->fst;
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  float adaptive_beam; int32_t best_idx;
  const double weight_cutoff = f_get_cutoff(d, l, n, &adaptive_beam, &best_idx);
  double next_cutoff = INFINITY;
  /* exact_cutoff == 3 (ordered admission, the lattice search's mode 3 for this decoder): the reference's online tightening of
   * next_weight_cutoff (faster-decoder.cc:266-273) as a prefix minimum over a canonical visiting order -- arcs leaving tokens whose
   * state falls into a lower one of 2^adm_bucket_bits buckets count as visited earlier -- starting from the bound the best token's
   * arcs give (:241-255).  pre[b]: what the arcs leaving a token of bucket b are admitted against (strictly below, :266). */
  const int ordered = d->cfg->exact_cutoff == 3;
  const int32_t bbits = d->cfg->adm_bucket_bits < 1 ? 1 : (d->cfg->adm_bucket_bits > 20 ? 20 : d->cfg->adm_bucket_bits);
#define VBO_FADM_BUCKET(state) ((uint32_t)((uint32_t)(state) * 2654435761u) >> (32 - bbits))
  double *pre = NULL, init_bound = INFINITY;
  if (ordered) {
    pre = (double *)malloc(sizeof(double) * ((size_t)1 << bbits));
    for (size_t b = 0; b < ((size_t)1 << bbits); b++) pre[b] = INFINITY;
    if (best_idx >= 0) {
      /* among equal-cost best tokens the smallest state id (the reference: the first in HashList order, :161-166) */
      const double bw = d->toks[l[best_idx].val].cost;
      for (int32_t i = 0; i < n; i++)
        if (d->toks[l[i].val].cost == bw && l[i].state < l[best_idx].state) best_idx = i;
      const int32_t state = l[best_idx].state, tok = l[best_idx].val;
      for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
        const vbo_arc *arc = &fst->arcs[a];
        if (arc->ilabel == 0) continue;
        const float ac_cost = -(d->acwt * d->loglikes[(size_t)frame * d->P + d->tid2pdf[arc->ilabel]]);
        const double new_weight = arc->weight + d->toks[tok].cost + ac_cost;
        if (new_weight + adaptive_beam < init_bound) init_bound = new_weight + adaptive_beam;
      }
    }
    next_cutoff = init_bound;
  }
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      fe_clear(f);
      if (ordered) {      /* exclusive prefix minimum over the buckets, from the initial bound */
        double run = init_bound;
        for (size_t b = 0; b < ((size_t)1 << bbits); b++) { const double m = pre[b]; pre[b] = run; if (m < run) run = m; }
      }
    }
    for (int32_t i = 0; i < n; i++) {
      const int32_t state = l[i].state, tok = l[i].val;
      if (!(d->toks[tok].cost < weight_cutoff)) continue;
      for (int64_t a = fst->arc_offset[state]; a < fst->arc_offset[state + 1]; a++) {
        const vbo_arc *arc = &fst->arcs[a];
        if (arc->ilabel == 0) continue;
        const float ac_cost = -(d->acwt * d->loglikes[(size_t)frame * d->P + d->tid2pdf[arc->ilabel]]);
        const double new_weight = arc->weight + d->toks[tok].cost + ac_cost;
        if (pass == 0) {
          if (new_weight + adaptive_beam < next_cutoff) next_cutoff = new_weight + adaptive_beam;
          if (ordered && new_weight + adaptive_beam < pre[VBO_FADM_BUCKET(state)]) pre[VBO_FADM_BUCKET(state)] = new_weight + adaptive_beam;
        }
        else if (ordered ? new_weight < pre[VBO_FADM_BUCKET(state)] : new_weight < next_cutoff) fe_relax(f, arc->nextstate, new_weight, a, tok, -1);
      }
    }
  }
#undef VBO_FADM_BUCKET
  free(pre);
  free(l);
  return next_cutoff;
}

int vbo_decode_faster(const vbo_fst *fst, const vbo_faster_cfg *cfg, const float *loglikes,
                      int32_t T, int32_t P, const int32_t *tid2pdf, int32_t num_tids,
                      float acwt, vbo_faster_result *res) {
  (void)num_tids;
  memset(res, 0, sizeof(*res));
  fdec dd; fdec *d = &dd;
  memset(d, 0, sizeof(*d));
  d->fst = fst; d->cfg = cfg; d->loglikes = loglikes; d->P = P; d->tid2pdf = tid2pdf; d->acwt = acwt;
  omap_init(&d->map, 1000);
  /* InitDecoding :36-45 */
  vbo_arc dummy = {0, 0, 0.0f, fst->start};
  int32_t nfd = 0;
  if (!cfg->exact_cutoff) {
    int32_t st = f_new_tok(d, &dummy, 0.0f, 0, -1);
    omap_insert(&d->map, fst->start, st);
    f_process_nonemitting(d, (double)FLT_MAX);
    while (nfd - 1 != T - 1) {
      double wc = f_process_emitting(d, nfd);
      nfd++;
      f_process_nonemitting(d, wc);
    }
  } else {
    fentries fe; fe_init(&fe);
    fe_relax(&fe, fst->start, 0.0, -1, -1, -1);
    fx_close_and_make(d, &fe, (double)FLT_MAX);
    while (nfd - 1 != T - 1) {
      const double wc = fx_process_emitting(d, &fe, nfd);
      nfd++;
      fx_close_and_make(d, &fe, wc);
    }
    fe_free(&fe);
  }
  res->num_frames = nfd;
  /* ReachedFinal :78-85, GetBestPath :87-147 */
  int32_t n; olist_item *l = omap_list(&d->map, &n);
  int is_final = 0;
  for (int32_t i = 0; i < n; i++)
    if (d->toks[l[i].val].cost != INFINITY && fst->final[l[i].state] != VBO_INF) is_final = 1;
  int32_t best = -1, best_state = -1;
  if (!is_final) {
    for (int32_t i = 0; i < n; i++) {
      const double c = d->toks[l[i].val].cost;
      if (best < 0 || d->toks[best].cost > c || (cfg->exact_cutoff && d->toks[best].cost == c && l[i].state < best_state)) { best = l[i].val; best_state = l[i].state; }
    }
  } else {
    double best_cost = INFINITY;
    for (int32_t i = 0; i < n; i++) {
      double c = d->toks[l[i].val].cost + fst->final[l[i].state];
      if (c == INFINITY) continue;
      if (c < best_cost || (cfg->exact_cutoff && c == best_cost && l[i].state < best_state)) { best_cost = c; best = l[i].val; best_state = l[i].state; }
    }
  }
  free(l);
  res->reached_final = is_final;
  res->ok = best >= 0;
  if (best >= 0) {
    int32_t len = 0;
    for (int32_t t = best; t >= 0; t = d->toks[t].prev) len++;
    res->ali = (int32_t *)malloc(sizeof(int32_t) * len); res->words = (int32_t *)malloc(sizeof(int32_t) * len);
    int32_t *chain = (int32_t *)malloc(sizeof(int32_t) * len);
    int32_t p = len;
    for (int32_t t = best; t >= 0; t = d->toks[t].prev) chain[--p] = t;
    float t1 = 0.0f, t2 = 0.0f;
    for (int32_t i = 1; i < len; i++) {  /* chain[0] is the dummy start token */
      const ftok *tk = &d->toks[chain[i]];
      float tot_cost = (float)(tk->cost - d->toks[tk->prev].cost), graph_cost = tk->weight,
            ac_cost = tot_cost - graph_cost;
      t1 = graph_cost + t1; t2 = ac_cost + t2;
      if (tk->ilabel != 0) res->ali[res->num_ali++] = tk->ilabel;
      if (tk->olabel != 0) res->words[res->num_words++] = tk->olabel;
    }
    if (is_final) t1 = fst->final[d->toks[best].nextstate] + t1;
    res->best_graph = t1; res->best_acoustic = t2;
    free(chain);
  }
  free(d->toks); free(d->queue); omap_free(&d->map);
  return 0;
}

void vbo_faster_result_free(vbo_faster_result *r) {
  free(r->ali); free(r->words);
  memset(r, 0, sizeof(*r));
}
