// C-ABI layer 3: rank-to-rank transport for the domain-decomposed solvers -- RCCL over xGMI, below the C ABI.
//
// Replaces the reference's MPI transport on the path: the MSG halo library
// (src/2d/ftn/mpi/mpi_msg.F:425-550: persistent MPI_Send_init/Recv_init channels, MPI_Start/MPI_Wait per
// exchange; driven from src/3d/mpi/msg_exchanger.cc:188-197 after every colour of
// src/3d/ftn/mpi/BMG3_SymStd_relax_GS.f90:102-147), the norm MPI_Allreduce (include/cedar/3d/mpi/grid_func.h:41)
// and the coarse-grid MPI_Allgatherv (include/cedar/3d/mpi/redist_solver.h:221-224).
//
// MI355X form: one communicator per process (= per GPU); a halo exchange is ONE ncclGroupStart/End bracket of
// ncclSend/ncclRecv to the neighbouring ranks (7 on the 2x2x2 node, each on its own xGMI link), enqueued on the
// library's current stream (cedar_amd_set_stream: the solver puts the y/z halo on a side stream under the interior
// rows), so the host never blocks inside a cycle.  librccl.so.1 is loaded on first use with RTLD_LOCAL: a process
// that never creates a communicator never maps it, and no torch / MPI is needed in a rank process -- the unique
// id travels as 128 opaque bytes from the launcher (cedar_amd/comm.py).
#include "../../include/cedar_amd.h"
#include "common.h"
#include "stage.h"
#include <dlfcn.h>
#include <cstring>
#include <cstdint>
#include <vector>
#include <rccl/rccl.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

using namespace cedar_amd;

namespace {

struct Rccl {
	void *h = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclAllReduce) AllReduce = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclBroadcast) Broadcast = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
	char why[256] = "";
};

Rccl &rccl()
{
	static Rccl r;
	static bool tried = false;
	if (tried) return r;
	tried = true;
	const char *names[] = { getenv("CEDAR_AMD_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so" };
	for (const char *n : names) {
		if (!n || !*n) continue;
		r.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (r.h) break;
		snprintf(r.why, sizeof(r.why), "%s", dlerror());
	}
	if (!r.h) return r;
#define SYM(f)                                                                   \
	r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.h, "nccl" #f));                \
	if (!r.f) {                                                                  \
		snprintf(r.why, sizeof(r.why), "librccl: symbol nccl" #f " missing");    \
		dlclose(r.h);                                                            \
		r.h = nullptr;                                                           \
		return r;                                                                \
	}
	SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(Send) SYM(Recv) SYM(AllReduce) SYM(AllGather)
	SYM(Broadcast) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
	return r;
}

int fail(const char *what, ncclResult_t e)
{
	char buf[256];
	snprintf(buf, sizeof(buf), "cedar_amd_comm: %s failed: %s", what, rccl().GetErrorString ? rccl().GetErrorString(e) : "?");
	print_error(buf);
	return 1;
}

#define NCCL_TRY(call, what)                      \
	do {                                          \
		ncclResult_t e_ = (call);                 \
		if (e_ != ncclSuccess) return fail(what, e_); \
	} while (0)

} // namespace

struct cedar_amd_comm {
	ncclComm_t c = nullptr;
	int rank = 0, world = 1;
};

extern "C" {

int cedar_amd_comm_available(void)
{
	if (rccl().h) return 1;
	return 0;
}

const char *cedar_amd_comm_why_unavailable(void) { return rccl().why; }

int cedar_amd_comm_unique_id(void *id128)
{
	if (!rccl().h) {
		char msg[] = "cedar_amd_comm_unique_id: librccl.so.1 could not be loaded";
		print_error(msg);
		return 1;
	}
	static_assert(sizeof(ncclUniqueId) == CEDAR_AMD_COMM_ID_BYTES, "unique id size");
	ncclUniqueId id;
	NCCL_TRY(rccl().GetUniqueId(&id), "ncclGetUniqueId");
	memcpy(id128, &id, sizeof(id));
	return 0;
}

cedar_amd_comm *cedar_amd_comm_create(const void *id128, int rank, int world)
{
	if (!rccl().h) {
		char msg[] = "cedar_amd_comm_create: librccl.so.1 could not be loaded";
		print_error(msg);
		return nullptr;
	}
	ncclUniqueId id;
	memcpy(&id, id128, sizeof(id));
	cedar_amd_comm *c = new cedar_amd_comm;
	c->rank = rank;
	c->world = world;
	ncclResult_t e = rccl().CommInitRank(&c->c, world, id, rank); // binds to the calling thread's current device
	if (e != ncclSuccess) {
		fail("ncclCommInitRank", e);
		delete c;
		return nullptr;
	}
	return c;
}

void cedar_amd_comm_destroy(cedar_amd_comm *c)
{
	if (!c) return;
	(void)hipDeviceSynchronize();
	if (c->c) (void)rccl().CommDestroy(c->c);
	delete c;
}

// ---- launcher-side bootstrap in compiled code: rank 0 makes the unique id and serves it over TCP next to MASTER_PORT, the
// other ranks fetch it (the protocol of cedar_amd/comm.py bootstrap_bytes, so that Python and C ranks of one job can
// mix: magic + 16-byte job tag (MASTER_PORT, world size, run id) + rank; the answer echoes magic + tag before the payload).
// Lets a plain C / C++ / Fortran host (Cedar's C interface, bmg_capi.cpp) create the communicator without Python or MPI.
namespace {

void sha1(const unsigned char *msg, size_t len, unsigned char out[20])
{
	uint32_t h0 = 0x67452301, h1 = 0xEFCDAB89, h2 = 0x98BADCFE, h3 = 0x10325476, h4 = 0xC3D2E1F0;
	const size_t total = ((len + 8) / 64 + 1) * 64;
	std::vector<unsigned char> m(total, 0);
	memcpy(m.data(), msg, len);
	m[len] = 0x80;
	const uint64_t bits = (uint64_t)len * 8;
	for (int i = 0; i < 8; i++) m[total - 1 - i] = (unsigned char)(bits >> (8 * i));
	for (size_t off = 0; off < total; off += 64) {
		uint32_t w[80];
		for (int i = 0; i < 16; i++)
			w[i] = (uint32_t)m[off + 4 * i] << 24 | (uint32_t)m[off + 4 * i + 1] << 16 | (uint32_t)m[off + 4 * i + 2] << 8 | m[off + 4 * i + 3];
		for (int i = 16; i < 80; i++) { uint32_t v = w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16]; w[i] = v << 1 | v >> 31; }
		uint32_t a = h0, b = h1, c = h2, d = h3, e = h4;
		for (int i = 0; i < 80; i++) {
			uint32_t f, k;
			if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999; }
			else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1; }
			else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
			else { f = b ^ c ^ d; k = 0xCA62C1D6; }
			const uint32_t t = (a << 5 | a >> 27) + f + e + k + w[i];
			e = d; d = c; c = b << 30 | b >> 2; b = a; a = t;
		}
		h0 += a; h1 += b; h2 += c; h3 += d; h4 += e;
	}
	const uint32_t hs[5] = {h0, h1, h2, h3, h4};
	for (int i = 0; i < 5; i++)
		for (int j = 0; j < 4; j++) out[4 * i + j] = (unsigned char)(hs[i] >> (24 - 8 * j));
}

bool send_all(int fd, const void *buf, size_t n)
{
	const char *p = static_cast<const char *>(buf);
	while (n) {
		const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
		if (k <= 0) return false;
		p += k; n -= (size_t)k;
	}
	return true;
}

bool recv_all(int fd, void *buf, size_t n)
{
	char *p = static_cast<char *>(buf);
	while (n) {
		const ssize_t k = ::recv(fd, p, n, 0);
		if (k <= 0) return false;
		p += k; n -= (size_t)k;
	}
	return true;
}

void set_timeout(int fd, int seconds)
{
	timeval tv{seconds, 0};
	setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
	setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

} // namespace

int cedar_amd_comm_bootstrap_id(void *id128, int rank, int world)
{
	if (world <= 1) return rank == 0 ? cedar_amd_comm_unique_id(id128) : 1;
	const char *host = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
	const char *mp = getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "29500";
	const int base = atoi(mp);
	const char *rid = getenv("CEDAR_AMD_RUN_ID") ? getenv("CEDAR_AMD_RUN_ID") : (getenv("TORCHELASTIC_RUN_ID") ? getenv("TORCHELASTIC_RUN_ID") : "");
	char key[256];
	const int klen = snprintf(key, sizeof(key), "%s:%d:%s", mp, world, rid);
	unsigned char dig[20];
	sha1(reinterpret_cast<const unsigned char *>(key), (size_t)klen, dig);
	unsigned char magic[28];
	memcpy(magic, "CEDARAMDUID1", 12);
	memcpy(magic + 12, dig, 16);
	sockaddr_in addr{};
	addr.sin_family = AF_INET;
	if (inet_pton(AF_INET, host, &addr.sin_addr) != 1) inet_pton(AF_INET, "127.0.0.1", &addr.sin_addr);
	if (rank == 0) {
		if (cedar_amd_comm_unique_id(id128)) return 1;
		int srv = -1;
		for (int i = 1; i <= 8 && srv < 0; i++) {
			const int fd = socket(AF_INET, SOCK_STREAM, 0);
			int one = 1;
			setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
			addr.sin_port = htons((uint16_t)(base + i));
			if (bind(fd, reinterpret_cast<sockaddr *>(&addr), sizeof(addr)) == 0 && listen(fd, world) == 0) srv = fd;
			else close(fd);
		}
		if (srv < 0) { char m[] = "cedar_amd_comm_bootstrap: no free port next to MASTER_PORT"; print_error(m); return 1; }
		set_timeout(srv, 120);
		std::vector<char> served((size_t)world, 0);
		int nserved = 0;
		while (nserved < world - 1) {
			const int c = accept(srv, nullptr, nullptr);
			if (c < 0) { close(srv); char m[] = "cedar_amd_comm_bootstrap: timed out waiting for the other ranks"; print_error(m); return 1; }
			set_timeout(c, 60);
			unsigned char got[28];
			uint32_t r = 0;
			if (recv_all(c, got, 28) && memcmp(got, magic, 28) == 0 && recv_all(c, &r, 4) && r >= 1 && r < (uint32_t)world && !served[r]) {
				const uint32_t n = CEDAR_AMD_COMM_ID_BYTES;
				if (send_all(c, magic, 28) && send_all(c, &n, 4) && send_all(c, id128, n)) { served[r] = 1; nserved++; }
			}
			close(c); // another job, a foreign client, a duplicate: neither answered nor counted
		}
		close(srv);
		return 0;
	}
	for (int attempt = 0; attempt < 1200; attempt++) { // ~120 s
		for (int i = 1; i <= 8; i++) {
			const int fd = socket(AF_INET, SOCK_STREAM, 0);
			addr.sin_port = htons((uint16_t)(base + i));
			set_timeout(fd, 3);
			if (connect(fd, reinterpret_cast<sockaddr *>(&addr), sizeof(addr)) == 0) {
				const uint32_t r = (uint32_t)rank;
				unsigned char got[28];
				uint32_t n = 0;
				if (send_all(fd, magic, 28) && send_all(fd, &r, 4) && recv_all(fd, got, 28) && memcmp(got, magic, 28) == 0
				    && recv_all(fd, &n, 4) && n == CEDAR_AMD_COMM_ID_BYTES && recv_all(fd, id128, n)) {
					close(fd);
					return 0;
				}
			}
			close(fd);
		}
		usleep(100000);
	}
	char m[] = "cedar_amd_comm_bootstrap: rank 0 did not serve the unique id";
	print_error(m);
	return 1;
}

cedar_amd_comm *cedar_amd_comm_bootstrap(int rank, int world)
{
	unsigned char id[CEDAR_AMD_COMM_ID_BYTES];
	if (cedar_amd_comm_bootstrap_id(id, rank, world)) return nullptr;
	return cedar_amd_comm_create(id, rank, world);
}

int cedar_amd_comm_rank(const cedar_amd_comm *c) { return c->rank; }
int cedar_amd_comm_size(const cedar_amd_comm *c) { return c->world; }

int cedar_amd_comm_exchange(cedar_amd_comm *c, int nsend, const int *speer, const real_t *const *sbuf, const size_t *scount,
                            int nrecv, const int *rpeer, real_t *const *rbuf, const size_t *rcount)
{
	if (nsend + nrecv == 0) return 0;
	hipStream_t st = current_stream();
	NCCL_TRY(rccl().GroupStart(), "ncclGroupStart");
	// receives first: the order inside a group does not matter to RCCL, but a self-message (rank talking to
	// itself in the one-GPU rehearsal) needs both halves in the same group anyway.  A failure inside the bracket
	// closes the group before returning: an open group would swallow every later RCCL call of the process.
	ncclResult_t e = ncclSuccess;
	const char *what = "";
	for (int i = 0; i < nrecv && e == ncclSuccess; i++)
		if (rcount[i]) { e = rccl().Recv(rbuf[i], rcount[i], ncclDouble, rpeer[i], c->c, st); what = "ncclRecv"; }
	for (int i = 0; i < nsend && e == ncclSuccess; i++)
		if (scount[i]) { e = rccl().Send(sbuf[i], scount[i], ncclDouble, speer[i], c->c, st); what = "ncclSend"; }
	if (e != ncclSuccess) {
		(void)rccl().GroupEnd();
		return fail(what, e);
	}
	NCCL_TRY(rccl().GroupEnd(), "ncclGroupEnd");
	return 0;
}

int cedar_amd_comm_allreduce_sum(cedar_amd_comm *c, real_t *buf, size_t n)
{
	NCCL_TRY(rccl().AllReduce(buf, buf, n, ncclDouble, ncclSum, c->c, current_stream()), "ncclAllReduce");
	return 0;
}

int cedar_amd_comm_allreduce_max(cedar_amd_comm *c, real_t *buf, size_t n)
{
	NCCL_TRY(rccl().AllReduce(buf, buf, n, ncclDouble, ncclMax, c->c, current_stream()), "ncclAllReduce");
	return 0;
}

int cedar_amd_comm_allgather(cedar_amd_comm *c, const real_t *send, real_t *recv, size_t count)
{
	NCCL_TRY(rccl().AllGather(send, recv, count, ncclDouble, c->c, current_stream()), "ncclAllGather");
	return 0;
}

int cedar_amd_comm_broadcast(cedar_amd_comm *c, real_t *buf, size_t count, int root)
{
	NCCL_TRY(rccl().Broadcast(buf, buf, count, ncclDouble, root, c->c, current_stream()), "ncclBroadcast");
	return 0;
}

// ---- streams (the side stream of the overlapped halo exchange; no torch in a rank process)
void *cedar_amd_stream_create(void)
{
	hipStream_t s = nullptr;
	CEDAR_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	return s;
}

void cedar_amd_stream_destroy(void *s)
{
	if (s) CEDAR_HIP_CHECK(hipStreamDestroy(static_cast<hipStream_t>(s)));
}

void cedar_amd_stream_wait(void *waiter, void *waited)
{
	hipEvent_t ev;
	CEDAR_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	CEDAR_HIP_CHECK(hipEventRecord(ev, static_cast<hipStream_t>(waited)));
	CEDAR_HIP_CHECK(hipStreamWaitEvent(static_cast<hipStream_t>(waiter), ev, 0));
	CEDAR_HIP_CHECK(hipEventDestroy(ev)); // released once the recorded work has completed
}

void cedar_amd_device_sync(void) { CEDAR_HIP_CHECK(hipDeviceSynchronize()); }
void cedar_amd_release_scratch(void)
{
	CEDAR_HIP_CHECK(hipDeviceSynchronize());
	cedar_amd::galerkin3_rows_release();
}

// ---- HIP events on the library's current stream (benchmarks time kernels with these, not with the host clock)
void *cedar_amd_event_record(void)
{
	hipEvent_t ev;
	CEDAR_HIP_CHECK(hipEventCreate(&ev));
	CEDAR_HIP_CHECK(hipEventRecord(ev, current_stream()));
	return ev;
}

float cedar_amd_event_elapsed_ms(void *e0, void *e1)
{
	float ms = 0;
	CEDAR_HIP_CHECK(hipEventSynchronize(static_cast<hipEvent_t>(e1)));
	CEDAR_HIP_CHECK(hipEventElapsedTime(&ms, static_cast<hipEvent_t>(e0), static_cast<hipEvent_t>(e1)));
	return ms;
}

void cedar_amd_event_destroy(void *e)
{
	if (e) CEDAR_HIP_CHECK(hipEventDestroy(static_cast<hipEvent_t>(e)));
}

} // extern "C"
