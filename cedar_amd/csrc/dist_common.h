// Shared pieces of the domain-decomposed solvers (dist3.cpp: 3D, dist2.cpp: 2D on (1, JJ, II) boxes): the rank's place in
// the grid, the transport (RCCL communicator or a caller-supplied table), the ghost-layer exchange and the side stream.
#pragma once
#include "../../include/cedar_amd.h"
#include "common.h"
#include "stage.h"
#include <cstring>
#include <map>
#include <utility>
#include <vector>

namespace cedar_amd {
namespace dist {

struct HaloEntry {
	int o[3];
	int peer;
	int sbox[6], rbox[6];
	size_t size, off;
};

struct HaloGroup {
	std::vector<int> idx;          // entries of the group
	std::vector<int> sboxes, rboxes; // 6 ints per entry
	std::vector<unsigned long long> offs;
};

struct Halo {
	int n[3] = {0, 0, 0};
	std::vector<HaloEntry> nb;
	size_t total = 0;
	HaloGroup grp[5]; // 0 = every neighbour, 1 = across an x face / edge / corner, 2 = the others (y/z),
	                  // 3 = the neighbours in the rank's own z layer (x / y faces and the four edges between them), 4 = the others
	std::map<long, std::pair<real_t *, real_t *>> bufs;
};

// what a rank knows about itself and how it talks to the others
struct RankCtx {
	cedar_amd_comm *comm = nullptr;
	cedar_amd_transport tp{};
	bool has_tp = false;
	int rank = 0, world = 1, p[3] = {1, 1, 1}, coord[3] = {0, 0, 0};
	void *side = nullptr; // non-blocking side stream of the overlapped y/z halo
	bool pending = false;
	real_t *scal = nullptr;
};

static inline real_t *dmalloc(size_t n)
{
	return static_cast<real_t *>(cedar_amd_malloc((n ? n : 1) * sizeof(real_t))); // cleared
}

// index range along one axis of extent n+2 for neighbour offset d (cedar_amd/dist.py _rng): d != 0: send = the owned
// layer next to that side, recv = the ghost layer; d == 0: the owned cells plus the ghost cell on every side that is a
// PHYSICAL boundary (those ghosts carry values the serial kernels compute for even extents)
static inline void rng(int d, int n, bool recv, bool has_minus, bool has_plus, int &lo, int &hi)
{
	if (d == 0) { lo = has_minus ? 1 : 0; hi = has_plus ? n + 1 : n + 2; }
	else if (d < 0) { lo = recv ? 0 : 1; hi = lo + 1; }
	else { lo = recv ? n + 1 : n; hi = lo + 1; }
}

static inline bool has_nb(const RankCtx *d, int dim, int side) { return d->coord[dim] + side >= 0 && d->coord[dim] + side < d->p[dim]; }
static inline int rank_of(const RankCtx *d, int ci, int cj, int ck) { return ci + d->p[0] * (cj + d->p[1] * ck); }

// ---- transport
static inline int tp_exchange(RankCtx *d, int ns, const int *speer, const real_t *const *sbuf, const size_t *scount,
                int nr, const int *rpeer, real_t *const *rbuf, const size_t *rcount)
{
	if (ns + nr == 0) return 0;
	if (d->has_tp) return d->tp.exchange(d->tp.ctx, ns, speer, sbuf, scount, nr, rpeer, rbuf, rcount);
	return cedar_amd_comm_exchange(d->comm, ns, speer, sbuf, scount, nr, rpeer, rbuf, rcount);
}

static inline void tp_allgather(RankCtx *d, const real_t *send, real_t *recv, size_t count)
{
	if (d->world == 1) {
		cedar_amd_memcpy_d2d(recv, send, count * sizeof(real_t));
		return;
	}
	const int rc = d->has_tp ? d->tp.allgather(d->tp.ctx, send, recv, count) : cedar_amd_comm_allgather(d->comm, send, recv, count);
	if (rc) { char m[] = "cedar_amd_dist3: all-gather failed"; print_error(m); }
}

static inline double tp_allreduce_sum(RankCtx *d, double v)
{
	if (d->world == 1) return v;
	if (d->has_tp) {
		if (d->tp.allreduce_sum(d->tp.ctx, &v, 1)) { char m[] = "cedar_amd_dist3: all-reduce failed"; print_error(m); }
		return v;
	}
	cedar_amd_memcpy_h2d(d->scal, &v, sizeof(double));
	if (cedar_amd_comm_allreduce_sum(d->comm, d->scal, 1)) { char m[] = "cedar_amd_dist3: all-reduce failed"; print_error(m); }
	cedar_amd_memcpy_d2h(&v, d->scal, sizeof(double));
	return v;
}

// ---- halo (cedar_amd/dist.py Halo)
static inline void halo_init(RankCtx *d, Halo &h, const int n[3])
{
	for (int t = 0; t < 3; t++) h.n[t] = n[t];
	bool hm[3], hp[3];
	for (int t = 0; t < 3; t++) { hm[t] = has_nb(d, t, -1); hp[t] = has_nb(d, t, +1); }
	size_t off = 0;
	// neighbours in the order of the sorted offsets (dx, dy, dz), as dist.py: both ends of a message agree on the layout
	for (int dx = -1; dx <= 1; dx++)
		for (int dy = -1; dy <= 1; dy++)
			for (int dz = -1; dz <= 1; dz++) {
				if (!dx && !dy && !dz) continue;
				const int c[3] = {d->coord[0] + dx, d->coord[1] + dy, d->coord[2] + dz};
				if (c[0] < 0 || c[0] >= d->p[0] || c[1] < 0 || c[1] >= d->p[1] || c[2] < 0 || c[2] >= d->p[2]) continue;
				HaloEntry e;
				e.o[0] = dx; e.o[1] = dy; e.o[2] = dz;
				e.peer = rank_of(d, c[0], c[1], c[2]);
				int lo[3], hi[3];
				for (int t = 0; t < 3; t++) rng(e.o[t], n[t], false, hm[t], hp[t], lo[t], hi[t]);
				for (int t = 0; t < 3; t++) { e.sbox[t] = lo[t]; e.sbox[3 + t] = hi[t] - lo[t]; }
				for (int t = 0; t < 3; t++) rng(e.o[t], n[t], true, hm[t], hp[t], lo[t], hi[t]);
				for (int t = 0; t < 3; t++) { e.rbox[t] = lo[t]; e.rbox[3 + t] = hi[t] - lo[t]; }
				e.size = (size_t)e.sbox[3] * e.sbox[4] * e.sbox[5];
				e.off = off;
				off += e.size;
				h.nb.push_back(e);
			}
	h.total = off;
	for (int g = 0; g < 5; g++) {
		HaloGroup &G = h.grp[g];
		for (size_t i = 0; i < h.nb.size(); i++) {
			const HaloEntry &e = h.nb[i];
			if ((g == 1 && e.o[0] == 0) || (g == 2 && e.o[0] != 0) || (g == 3 && e.o[2] != 0) || (g == 4 && e.o[2] == 0)) continue;
			G.idx.push_back((int)i);
			for (int t = 0; t < 6; t++) { G.sboxes.push_back(e.sbox[t]); G.rboxes.push_back(e.rbox[t]); }
			G.offs.push_back((unsigned long long)e.off);
		}
	}
}

static inline std::pair<real_t *, real_t *> &halo_bufs(Halo &h, long key, size_t count)
{
	auto it = h.bufs.find(key);
	if (it == h.bufs.end()) it = h.bufs.emplace(key, std::make_pair(dmalloc(count), dmalloc(count))).first;
	return it->second;
}

// fill every ghost cell owned by a neighbour of the group: pack (one launch) -> one grouped exchange -> unpack
static inline void halo_exchange(RankCtx *d, Halo &h, int II, int JJ, int KK, real_t *arr, int nplanes, int group)
{
	HaloGroup &G = h.grp[group];
	if (G.idx.empty()) return;
	auto &bp = halo_bufs(h, nplanes, h.total * (size_t)nplanes);
	real_t *sb = bp.first, *rb = bp.second;
	const int nbx = (int)G.idx.size();
	cedar_amd_box_copy(arr, II, JJ, KK, nplanes, nbx, G.sboxes.data(), G.offs.data(), sb, 0);
	int peer[26];
	const real_t *sp[26];
	real_t *rp[26];
	size_t cnt[26];
	for (int i = 0; i < nbx; i++) {
		const HaloEntry &e = h.nb[G.idx[i]];
		peer[i] = e.peer;
		sp[i] = sb + e.off * (size_t)nplanes;
		rp[i] = rb + e.off * (size_t)nplanes;
		cnt[i] = e.size * (size_t)nplanes;
	}
	if (tp_exchange(d, nbx, peer, sp, cnt, nbx, peer, rp, cnt)) { char m[] = "cedar_amd_dist3: halo exchange failed"; print_error(m); }
	cedar_amd_box_copy(arr, II, JJ, KK, nplanes, nbx, G.rboxes.data(), G.offs.data(), rb, 1);
}

// The ghost cells of this rank's own z layer / across z that one stage of the boundary-first chain has changed: of the
// messages selected by `send` / `recv` (neighbour offsets) only the rows with index parity jpar_of(offset) and the planes
// with index parity kpar (-1: all).  Local extents are even along split directions, so a row or plane has the same parity on both
// ends of a message and both ends find the same boxes (an empty one is no message).
template <class FS, class FR, class FJ>
static inline void halo_exchange_sub(RankCtx *d, Halo &h, int II, int JJ, int KK, real_t *arr, FS send, FR recv, FJ jpar_of, int kpar)
{
	if (h.nb.empty()) return;
	auto &bp = halo_bufs(h, 1, h.total);
	real_t *sb = bp.first, *rb = bp.second;
	int sboxes[26 * 8], rboxes[26 * 8], speer[26], rpeer[26], ns = 0, nr = 0;
	unsigned long long soffs[26], roffs[26];
	const real_t *sp[26];
	real_t *rp[26];
	size_t scnt[26], rcnt[26];
	auto restrict_box = [&](const int *box, int *out, int jpar) -> size_t {
		int j0 = box[1], nj = box[4], sj = 1, k0 = box[2], nk = box[5], sk = 1;
		if (jpar >= 0) { const int f = ((j0 & 1) == jpar) ? j0 : j0 + 1; nj = (j0 + nj - f + 1) / 2; j0 = f; sj = 2; }
		if (kpar >= 0) { const int f = ((k0 & 1) == kpar) ? k0 : k0 + 1; nk = (k0 + nk - f + 1) / 2; k0 = f; sk = 2; }
		if (nj < 0) nj = 0;
		if (nk < 0) nk = 0;
		out[0] = box[0]; out[1] = j0; out[2] = k0; out[3] = box[3]; out[4] = nj; out[5] = nk; out[6] = sj; out[7] = sk;
		return (size_t)box[3] * nj * nk;
	};
	for (const HaloEntry &e : h.nb) {
		if (send(e.o)) {
			const size_t c = restrict_box(e.sbox, sboxes + 8 * ns, jpar_of(e.o));
			if (c) { speer[ns] = e.peer; soffs[ns] = e.off; sp[ns] = sb + e.off; scnt[ns] = c; ns++; }
		}
		if (recv(e.o)) {
			const size_t c = restrict_box(e.rbox, rboxes + 8 * nr, jpar_of(e.o));
			if (c) { rpeer[nr] = e.peer; roffs[nr] = e.off; rp[nr] = rb + e.off; rcnt[nr] = c; nr++; }
		}
	}
	if (ns) cedar_amd_box_copy_strided(arr, II, JJ, KK, 1, ns, sboxes, soffs, sb, 0);
	if (tp_exchange(d, ns, speer, sp, scnt, nr, rpeer, rp, rcnt)) { char m[] = "cedar_amd_dist3: halo exchange failed"; print_error(m); }
	if (nr) cedar_amd_box_copy_strided(arr, II, JJ, KK, 1, nr, rboxes, roffs, rb, 1);
}

// x faces only (owned j,k): to_minus: first owned column to the -x neighbour, the +x neighbour's into the high ghost
// column (UP order); else the mirror image.  Returns true if a column was received.
static inline bool halo_exchange_x(RankCtx *d, Halo &h, int II, int JJ, int KK, real_t *arr, bool to_minus)
{
	const int nx = h.n[0], ny = h.n[1], nz = h.n[2] < 1 ? 1 : h.n[2], k0 = h.n[2] < 1 ? 0 : 1; // 2D boxes: the one plane k = 0
	const int send_to = to_minus ? -1 : +1, send_col = to_minus ? 1 : nx, recv_from = -send_to, recv_col = to_minus ? nx + 1 : 0;
	auto &bp = halo_bufs(h, to_minus ? -1 : -2, (size_t)ny * nz);
	const unsigned long long zero = 0;
	int speer = 0, rpeer = 0, ns = 0, nr = 0;
	const real_t *sp = bp.first;
	real_t *rp = bp.second;
	size_t cnt = (size_t)ny * nz;
	if (has_nb(d, 0, send_to)) {
		const int box[6] = {send_col, 1, k0, 1, ny, nz};
		cedar_amd_box_copy(arr, II, JJ, KK, 1, 1, box, &zero, bp.first, 0);
		speer = rank_of(d, d->coord[0] + send_to, d->coord[1], d->coord[2]);
		ns = 1;
	}
	if (has_nb(d, 0, recv_from)) {
		rpeer = rank_of(d, d->coord[0] + recv_from, d->coord[1], d->coord[2]);
		nr = 1;
	}
	if (tp_exchange(d, ns, &speer, &sp, &cnt, nr, &rpeer, &rp, &cnt)) { char m[] = "cedar_amd_dist3: x-face exchange failed"; print_error(m); }
	if (nr) {
		const int box[6] = {recv_col, 1, k0, 1, ny, nz};
		cedar_amd_box_copy(arr, II, JJ, KK, 1, 1, box, &zero, bp.second, 1);
	}
	return nr != 0;
}

// ---- side stream: work issued between side_begin / side_end goes to the side stream, ordered after everything
// already queued on the main stream; side_wait orders the main stream after it
static inline void *side_begin(RankCtx *d)
{
	if (!d->side) d->side = cedar_amd_stream_create();
	void *main_st = cedar_amd_get_stream();
	cedar_amd_stream_wait(d->side, main_st);
	cedar_amd_set_stream(d->side);
	return main_st;
}
static inline void side_end(RankCtx *d, void *main_st)
{
	cedar_amd_set_stream(main_st);
	d->pending = true;
}
static inline void side_wait(RankCtx *d)
{
	if (!d->pending) return;
	cedar_amd_stream_wait(cedar_amd_get_stream(), d->side);
	d->pending = false;
}


} // namespace dist
} // namespace cedar_amd
