/*
 * rccl_double.c - TEST INFRASTRUCTURE, never part of the product: a stand-in for the RCCL entry points libvictor_hip.so
 * looks up with dlsym (victor_amd/csrc/vk_rccl.cpp: open_rccl), selected through VICTOR_HIP_RCCL_LIB.
 *
 * Why: a one-GPU box cannot build an RCCL communicator of two ranks (RCCL wants one device per rank), so the N > 1 branch of
 * vk_comm_init / vk_comm_allgather_async - the 128-byte id crossing processes by value, the rank-major receive layout, the
 * padded last shard - never ran.  This double gives ranks that SHARE a device a working communicator: it moves the bytes
 * with hipMemcpy and a Unix-domain socket (star topology, rank 0 at the centre) instead of xGMI.  Same prototypes as rccl.h
 * (ncclUniqueId is 128 bytes passed BY VALUE; ncclFloat64 = 8); ncclAllGather synchronises the stream and is blocking, which
 * a caller written for the asynchronous original cannot tell.  Says nothing about RCCL's performance and is never timed.
 *
 * Build (tests/test_gpu_rccl_double.py does it): gcc -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_double.c
 *                                                -L/opt/rocm/lib -lamdhip64 -o librccl_double.so
 */
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#define ID_BYTES 128
typedef struct { char internal[ID_BYTES]; } ncclUniqueId;
typedef struct comm {
  int rank, nranks;
  int root_fd;   /* ranks > 0: connection to rank 0 */
  int* peer_fd;  /* rank 0: connections of ranks 1..n-1, by rank */
  int listen_fd;
  char path[108];
  long allgathers;
} comm_t;

enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };

static int send_all(int fd, const void* buf, size_t n) {
  const char* p = (const char*)buf;
  while (n) {
    ssize_t k = send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return -1;
    }
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

static int recv_all(int fd, void* buf, size_t n) {
  char* p = (char*)buf;
  while (n) {
    ssize_t k = recv(fd, p, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return -1;
    }
    p += k;
    n -= (size_t)k;
  }
  return 0;
}

const char* ncclGetErrorString(int code) {
  switch (code) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl_double: a HIP call failed";
    case ncclSystemError: return "rccl_double: socket error";
    case ncclInvalidArgument: return "rccl_double: invalid argument";
    case ncclInvalidUsage: return "rccl_double: not provided by the test double";
    default: return "rccl_double: internal error";
  }
}

int ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 1;          /* not an RCCL version: the JSON line of a run on this double says so */
  return ncclSuccess;
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  const char* dir = getenv("RCCL_DOUBLE_DIR");
  snprintf(id->internal, ID_BYTES, "%s/rccl_double_%d_%ld%09ld.sock", dir ? dir : "/tmp", (int)getpid(), (long)ts.tv_sec, ts.tv_nsec);
  return ncclSuccess;
}

int ncclCommInitRank(comm_t** out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  id.internal[ID_BYTES - 1] = 0;
  comm_t* c = (comm_t*)calloc(1, sizeof *c);
  if (!c) return ncclInternalError;
  c->rank = rank;
  c->nranks = nranks;
  c->root_fd = c->listen_fd = -1;
  snprintf(c->path, sizeof c->path, "%.100s", id.internal);
  struct sockaddr_un addr;
  memset(&addr, 0, sizeof addr);
  addr.sun_family = AF_UNIX;
  snprintf(addr.sun_path, sizeof addr.sun_path, "%s", c->path);
  if (nranks > 1 && rank == 0) {
    c->peer_fd = (int*)calloc((size_t)nranks, sizeof(int));
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    unlink(c->path);
    if (c->listen_fd < 0 || bind(c->listen_fd, (struct sockaddr*)&addr, sizeof addr) || listen(c->listen_fd, nranks)) return ncclSystemError;
    for (int k = 1; k < nranks; ++k) {
      int fd = accept(c->listen_fd, NULL, NULL);
      int32_t peer = -1;
      if (fd < 0 || recv_all(fd, &peer, sizeof peer) || peer < 1 || peer >= nranks || c->peer_fd[peer]) return ncclSystemError;
      c->peer_fd[peer] = fd;
    }
    for (int k = 1; k < nranks; ++k) {
      int32_t ok = nranks;
      if (send_all(c->peer_fd[k], &ok, sizeof ok)) return ncclSystemError;
    }
  } else if (nranks > 1) {
    int fd = -1;
    for (int attempt = 0; attempt < 3000; ++attempt) {          /* rank 0 may not be listening yet: up to 60 s */
      fd = socket(AF_UNIX, SOCK_STREAM, 0);
      if (fd >= 0 && connect(fd, (struct sockaddr*)&addr, sizeof addr) == 0) break;
      if (fd >= 0) close(fd);
      fd = -1;
      struct timespec nap = {0, 20000000};
      nanosleep(&nap, NULL);
    }
    int32_t me = rank, ok = 0;
    if (fd < 0 || send_all(fd, &me, sizeof me) || recv_all(fd, &ok, sizeof ok) || ok != nranks) return ncclSystemError;
    c->root_fd = fd;
  }
  *out = c;
  return ncclSuccess;
}

/* every rank contributes count elements at sendbuf (device), every rank receives nranks * count at recvbuf (device),
 * rank r's part at offset r * count - the layout of ncclAllGather */
int ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, int datatype, comm_t* c, hipStream_t stream) {
  if (!c || !sendbuf || !recvbuf) return ncclInvalidArgument;
  if (datatype != 8 && datatype != 7) return ncclInvalidArgument;            /* ncclFloat64 (8) / ncclFloat32 (7) */
  const size_t bytes = count * (datatype == 8 ? 8 : 4);
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;   /* what the stream computed is the input */
  char* all = (char*)malloc(bytes * (size_t)c->nranks + 1);
  if (!all) return ncclInternalError;
  int rc = ncclSuccess;
  if (hipMemcpy(all + bytes * (size_t)c->rank, sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
  if (rc == ncclSuccess && c->nranks > 1) {
    if (c->rank == 0) {
      for (int k = 1; k < c->nranks && rc == ncclSuccess; ++k) {
        uint64_t theirs = 0;
        if (recv_all(c->peer_fd[k], &theirs, sizeof theirs) || theirs != bytes || recv_all(c->peer_fd[k], all + bytes * (size_t)k, bytes))
          rc = ncclSystemError;            /* a rank that disagrees about the count: an error, not a hang */
      }
      for (int k = 1; k < c->nranks && rc == ncclSuccess; ++k)
        if (send_all(c->peer_fd[k], all, bytes * (size_t)c->nranks)) rc = ncclSystemError;
    } else {
      uint64_t mine = bytes;
      if (send_all(c->root_fd, &mine, sizeof mine) || send_all(c->root_fd, all + bytes * (size_t)c->rank, bytes) ||
          recv_all(c->root_fd, all, bytes * (size_t)c->nranks))
        rc = ncclSystemError;
    }
  }
  if (rc == ncclSuccess && hipMemcpy(recvbuf, all, bytes * (size_t)c->nranks, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
  free(all);
  c->allgathers += 1;
  return rc;
}

int ncclCommDestroy(comm_t* c) {
  if (!c) return ncclSuccess;
  if (c->root_fd >= 0) close(c->root_fd);
  if (c->peer_fd) {
    for (int k = 1; k < c->nranks; ++k)
      if (c->peer_fd[k] > 0) close(c->peer_fd[k]);
    free(c->peer_fd);
  }
  if (c->listen_fd >= 0) {
    close(c->listen_fd);
    unlink(c->path);
  }
  free(c);
  return ncclSuccess;
}

/* the one-process layout (ncclCommInitAll + grouped calls) is exercised against the real RCCL; the double refuses it */
int ncclCommInitAll(comm_t** comms, int ndev, const int* devlist) {
  (void)comms; (void)ndev; (void)devlist;
  return ncclInvalidUsage;
}
/* what a communicator says about itself (vk_comm_rank_info: a multi-GPU record carries these per rank) */
int ncclCommCount(const comm_t* c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->nranks;
  return ncclSuccess;
}

int ncclCommUserRank(const comm_t* c, int* rank) {
  if (!c || !rank) return ncclInvalidArgument;
  *rank = c->rank;
  return ncclSuccess;
}

int ncclCommCuDevice(const comm_t* c, int* device) {
  if (!c || !device) return ncclInvalidArgument;
  return hipGetDevice(device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;      /* (ranks of the double share a device) */
}

int ncclGroupStart(void) { return ncclSuccess; }
int ncclGroupEnd(void) { return ncclSuccess; }
