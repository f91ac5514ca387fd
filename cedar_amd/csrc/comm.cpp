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
#include <rccl/rccl.h>

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
