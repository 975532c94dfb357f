// vk_ledger.cpp - the polling hand-off's launch rule and the device-wide ledger of reserved waiters (vk_ledger.h).  Host code
// only (POSIX, no HIP): compiled by the host compiler next to the kernels' translation units and linked into libvictor_hip.so.
//
// The per-process budget bounds ONE process; the bound that excludes a deadlock is device-wide (fewer than kPollXcdSlots waiters
// resident on a GPU).  Processes of one user on one host therefore keep their reservations in a small file in /dev/shm, one per
// GPU (named after its PCI bus id): a slot per owner, written by that owner only; a reservation is granted when the slots of
// the LIVING owners, the new one included, stay below the bound (optimistic: publish, re-read the sum, take it back if two
// raced past the bound).  Round 6 hardened it (VERDICT / ADVICE round 5):
//   * an owner is {pid, start time, pid-namespace inode, library instance}, not a pid: a recycled pid does not inherit a dead
//     process's reservation, a small container pid does not adopt a stranger's slot, two copies of the library in one process
//     keep a slot each;
//   * a slot written in ANOTHER pid namespace (containers sharing /dev/shm) cannot be judged from here - kill() and /proc speak
//     about our namespace's numbers -, so it counts as living and is never taken over: the bound errs towards the counters;
//   * the file is opened without following a symbolic link and must be a regular file of this user, closed to group and others,
//     of exactly the expected size: anything else and the process does not poll at all (kUntrusted);
//   * slots change hands under an advisory lock on the file (claims are rare: once per process and GPU); the counts themselves
//     stay lock-free, with sequentially consistent publish-then-check so that of two racing owners at least one sees the other.

#include "vk_ledger.h"

#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>

#include "victor_hip.h"

namespace vkl {

namespace {

constexpr int kLedgerSlots = 126;
constexpr uint32_t kLedgerMagic = 0x564b504cu;   // "VKPL"
constexpr uint32_t kLedgerVersion = 2;           // version 1 (round 5: {pid, reserved} slots) lived in victor_hip_poll_<uid>_<bus>

struct Slot {
  std::atomic<int64_t> pid;        // 0: free.  Published last by a claim, cleared first by a take-over (readers re-check it)
  std::atomic<uint64_t> start, ns, lib;
  std::atomic<int32_t> reserved;
  int32_t pad;
};
struct File {
  uint32_t magic, version;
  std::atomic<uint32_t> gen;
  uint32_t pad;
  Slot slot[kLedgerSlots];
};
static_assert(sizeof(Slot) == 40 && sizeof(File) == 16 + 40 * kLedgerSlots, "ledger layout (mirrored in tests/test_ledger.py)");
static_assert(std::atomic<int64_t>::is_always_lock_free && std::atomic<int32_t>::is_always_lock_free, "slots are shared between processes");

std::atomic<uint32_t> g_process_gen{0};      // releases of this library instance (the no-ledger case has no file to count them in)
const int g_lib_anchor = 0;                  // its address names this copy of the library inside the process

uint64_t pid_namespace_inode() {
  struct stat st;
  return stat("/proc/self/ns/pid", &st) == 0 ? (uint64_t)st.st_ino : 0;
}

}  // namespace

struct Ledger {
  File* map = nullptr;
  int mine = -1;
  Identity me;
};

int process_state(int64_t pid, uint64_t* start) {
  if (start) *start = 0;
  if (pid <= 0) return 0;
  if (kill((pid_t)pid, 0) != 0 && errno == ESRCH) return 0;          // (EPERM: somebody else's process, but a process)
  // kill(pid, 0) answers for zombies as well (a dead child nobody has waited for yet): the state letter decides for those, and
  // field 22 is the start time that tells a recycled pid from the process that wrote the slot
  char path[64], buf[1024];
  snprintf(path, sizeof path, "/proc/%lld/stat", (long long)pid);
  FILE* f = fopen(path, "r");
  if (!f) return 1;                           // no /proc here: kill() has spoken
  const size_t got = fread(buf, 1, sizeof buf - 1, f);
  fclose(f);
  buf[got] = 0;
  const char* p = strrchr(buf, ')');          // the command name may contain anything, also ')'
  if (!p || p[1] != ' ' || !p[2]) return 1;
  const char state = p[2];
  if (start) {
    // p + 2 is field 3 (state); field 22 is 19 fields on
    const char* q = p + 2;
    for (int field = 3; field < 22 && q; ++field) {
      q = strchr(q, ' ');
      if (q) ++q;
    }
    if (q) *start = strtoull(q, nullptr, 10);
  }
  return (state == 'Z' || state == 'X') ? 0 : 1;
}

Identity self_identity() {
  Identity me;
  me.pid = (int64_t)getpid();
  (void)process_state(me.pid, &me.start);
  me.ns = pid_namespace_inode();
  me.lib = (uint64_t)(uintptr_t)&g_lib_anchor;
  return me;
}

namespace {

// Is the owner of a slot still there, as far as `me` can tell?  Another namespace: cannot be told - yes.
bool owner_present(const Identity& me, int64_t pid, uint64_t start, uint64_t ns) {
  if (ns != me.ns) return true;
  uint64_t now_start = 0;
  if (process_state(pid, &now_start) == 0) return false;
  if (start != 0 && now_start != 0 && start != now_start) return false;      // the pid names another process by now
  return true;
}

struct SlotView { int64_t pid; uint64_t start, ns, lib; int32_t reserved; };

// a consistent view of a slot (its owner fields may be rewritten by a take-over while we read): false = free
bool read_slot(const Slot& s, SlotView* v) {
  for (int attempt = 0; attempt < 4; ++attempt) {
    v->pid = s.pid.load(std::memory_order_seq_cst);
    if (v->pid == 0) return false;
    v->start = s.start.load(std::memory_order_acquire);
    v->ns = s.ns.load(std::memory_order_acquire);
    v->lib = s.lib.load(std::memory_order_acquire);
    v->reserved = s.reserved.load(std::memory_order_seq_cst);
    if (s.pid.load(std::memory_order_seq_cst) == v->pid) return true;
  }
  v->ns = ~uint64_t(0);        // changing hands again and again: count what was read as a stranger's (conservative)
  return true;
}

bool same_owner(const SlotView& v, const Identity& me) { return v.pid == me.pid && v.start == me.start && v.ns == me.ns && v.lib == me.lib; }

void write_owner(Slot& s, const Identity& me) {
  s.reserved.store(0, std::memory_order_seq_cst);
  s.start.store(me.start, std::memory_order_release);
  s.ns.store(me.ns, std::memory_order_release);
  s.lib.store(me.lib, std::memory_order_release);
  s.pid.store(me.pid, std::memory_order_seq_cst);
}

bool lock_file(int fd) {
  for (int attempt = 0; attempt < 200; ++attempt) {          // claims take microseconds; 200 ms without the lock: somebody sits on it
    if (flock(fd, LOCK_EX | LOCK_NB) == 0) return true;
    if (errno != EWOULDBLOCK && errno != EINTR) return false;
    struct timespec ts = {0, 1000000};
    nanosleep(&ts, nullptr);
  }
  return false;
}

}  // namespace

Ledger* open_at(const char* path, const Identity& me, int* status) {
  int st_dummy;
  if (!status) status = &st_dummy;
  *status = kUnavailable;
  if (!path || me.pid <= 0) return nullptr;
  const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC | O_NOFOLLOW, 0600);
  if (fd < 0) {
    struct stat ls;
    // something IS there and we may not use it (a symbolic link: ELOOP; another user's file: EACCES): not ours to trust.
    // Nothing there and no way to create it (no /dev/shm, read-only): unavailable.
    *status = (errno == ELOOP || lstat(path, &ls) == 0) ? kUntrusted : kUnavailable;
    return nullptr;
  }
  Ledger* led = nullptr;
  struct stat st;
  do {
    if (fstat(fd, &st) != 0) break;
    if (!S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 0077) != 0 || st.st_nlink != 1) {
      *status = kUntrusted;
      break;
    }
    if (!lock_file(fd)) {
      *status = kUntrusted;
      break;
    }
    if (fstat(fd, &st) != 0) break;                                   // (the size, now that nobody else is stamping it)
    if (st.st_size == 0) {
      if (ftruncate(fd, sizeof(File)) != 0) break;
    } else if (st.st_size != (off_t)sizeof(File)) {
      *status = kUntrusted;
      break;
    }
    void* m = mmap(nullptr, sizeof(File), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) break;
    File* file = static_cast<File*>(m);
    if (file->magic == 0 && file->version == 0) {      // fresh (zero-filled): stamped under the lock
      file->version = kLedgerVersion;
      file->magic = kLedgerMagic;
    }
    if (file->magic != kLedgerMagic || file->version != kLedgerVersion) {
      munmap(m, sizeof(File));
      *status = kUntrusted;
      break;
    }
    // the claim, under the lock: my own slot (adopted as it is), else a free one, else one whose owner is gone
    int mine = -1, free_slot = -1, stale = -1;
    for (int i = 0; i < kLedgerSlots && mine < 0; ++i) {
      SlotView v;
      if (!read_slot(file->slot[i], &v)) {
        if (free_slot < 0) free_slot = i;
      } else if (same_owner(v, me)) {
        mine = i;
      } else if (stale < 0 && !owner_present(me, v.pid, v.start, v.ns)) {
        stale = i;
      }
    }
    if (mine < 0) {
      mine = free_slot >= 0 ? free_slot : stale;
      if (mine >= 0) {
        Slot& s = file->slot[mine];
        s.pid.store(0, std::memory_order_seq_cst);      // (a reader that saw the old owner re-checks the pid and starts over)
        write_owner(s, me);
        file->gen.fetch_add(1, std::memory_order_seq_cst);
      }
    }
    if (mine < 0) {
      munmap(m, sizeof(File));
      *status = kFull;
      break;
    }
    led = new (std::nothrow) Ledger();
    if (!led) {
      munmap(m, sizeof(File));
      break;
    }
    led->map = file;
    led->mine = mine;
    led->me = me;
    *status = kOpened;
  } while (false);
  (void)flock(fd, LOCK_UN);
  ::close(fd);
  return led;
}

void close(Ledger* led, bool keep_slot) {
  if (!led) return;
  if (!keep_slot) {
    Slot& s = led->map->slot[led->mine];
    s.reserved.store(0, std::memory_order_seq_cst);
    s.pid.store(0, std::memory_order_seq_cst);
    led->map->gen.fetch_add(1, std::memory_order_seq_cst);
  }
  munmap(led->map, sizeof(File));
  delete led;
}

int slot_of(const Ledger* led) { return led ? led->mine : -1; }

int others(const Ledger* led) {
  int total = 0;
  for (int i = 0; i < kLedgerSlots; ++i) {
    if (i == led->mine) continue;
    SlotView v;
    if (!read_slot(led->map->slot[i], &v)) continue;
    if (owner_present(led->me, v.pid, v.start, v.ns)) total += std::max(0, (int)v.reserved);
  }
  return total;
}

uint32_t generation(const Ledger* led) {
  return g_process_gen.load(std::memory_order_relaxed) + (led ? led->map->gen.load(std::memory_order_relaxed) : 0u);
}

int grant(Ledger* led, std::atomic<int>* process, int ctx_reserved, int want) {
  int seen = process->load(std::memory_order_relaxed), g;
  do {
    g = vk_poll_grant(led ? others(led) : 0, seen, ctx_reserved, want);
  } while (g > 0 && !process->compare_exchange_weak(seen, seen + g, std::memory_order_relaxed));
  if (g > 0 && led) {
    // published; if another owner raced past the bound meanwhile, take it back (both may: conservative)
    Slot& s = led->map->slot[led->mine];
    s.reserved.fetch_add(g, std::memory_order_seq_cst);
    if (others(led) + process->load(std::memory_order_relaxed) >= kPollXcdSlots) {
      s.reserved.fetch_sub(g, std::memory_order_seq_cst);
      process->fetch_sub(g, std::memory_order_relaxed);
      g = 0;
    }
  }
  return g > 0 ? g : 0;
}

void release(Ledger* led, std::atomic<int>* process, int n) {
  if (n <= 0) return;
  process->fetch_sub(n, std::memory_order_relaxed);
  g_process_gen.fetch_add(1, std::memory_order_relaxed);
  if (led) {
    led->map->slot[led->mine].reserved.fetch_sub(n, std::memory_order_seq_cst);
    led->map->gen.fetch_add(1, std::memory_order_seq_cst);
  }
}

Ledger* for_device(const std::string& bus, int* status) {
  struct Kept { Ledger* led; int status; };
  static std::mutex mu;                                  // (the serving threads of an owner process launch concurrently)
  static std::map<std::string, Kept> kept;
  std::lock_guard<std::mutex> hold(mu);
  auto it = kept.find(bus);
  if (it == kept.end()) {
    std::string name = "/dev/shm/victor_hip_poll2_" + std::to_string((unsigned)geteuid()) + "_";
    for (char c : bus) name += (isalnum((unsigned char)c) ? c : '_');
    Kept k{nullptr, kUnavailable};
    k.led = open_at(name.c_str(), self_identity(), &k.status);
    it = kept.emplace(bus, k).first;
  }
  if (status) *status = it->second.status;
  return it->second.led;
}

}  // namespace vkl

// ---- the launch rule of the polling hand-off: pure functions of their arguments (include/victor_hip.h) --------------------
extern "C" {

int32_t vk_poll_rule(int64_t n_points, int32_t parts, int64_t workgroups, int32_t workgroups_per_cu, int32_t n_cu, int32_t reserved) {
  if (parts < 2 || n_points < 1 || n_points > vkl::kPollPoints) return 0;      // nothing to hand over / more waiters than a launch may hold
  if (workgroups_per_cu < 2 || n_cu < 8) return 0;                         // an XCD must offer kPollXcdSlots = 32 CUs x 2 slots at least
  if ((long long)(n_cu / 8) * workgroups_per_cu < vkl::kPollXcdSlots) return 0;
  if (workgroups > (long long)workgroups_per_cu * n_cu) return 0;          // the launch must fit on the chip at once
  if (n_points > reserved) return 0;                                       // its waiters must be covered by the context's reservation
  return 1;
}

int32_t vk_poll_grant(int32_t others_reserved, int32_t process_reserved, int32_t ctx_reserved, int32_t want) {
  if (want > (int32_t)vkl::kPollPoints) want = (int32_t)vkl::kPollPoints;
  if (want <= ctx_reserved || process_reserved < 0 || others_reserved < 0) return 0;
  const int32_t extra = want - ctx_reserved;
  const int32_t room = std::min(vkl::kPollBudget - process_reserved,                           // this process's budget
                                vkl::kPollXcdSlots - 1 - others_reserved - process_reserved);  // fewer than an XCD's slots on the device
  return extra <= room ? extra : 0;      // all or nothing: a launch polls for every one of its points or for none
}

int32_t vk_poll_budget(int32_t* per_process, int32_t* xcd_slots) {
  if (per_process) *per_process = vkl::kPollBudget;
  if (xcd_slots) *xcd_slots = vkl::kPollXcdSlots;
  return (vkl::kPollXcdSlots - 1) / vkl::kPollBudget;       // processes per device that may each use their full budget
}

// ---- the ledger's operations on a file of the caller's choosing, with an owner of the caller's making: what tests/test_ledger.py
// drives without a GPU.  Development entry points: they answer only with VICTOR_HIP_DEV=1 in the environment, like every other
// development switch of the library.
static bool ledger_dev() {
  const char* dev = getenv("VICTOR_HIP_DEV");
  return dev && strcmp(dev, "1") == 0;
}

void vk_ledger_layout(int32_t* header_bytes, int32_t* slot_bytes, int32_t* slots, int32_t* version) {
  if (header_bytes) *header_bytes = 16;
  if (slot_bytes) *slot_bytes = (int32_t)sizeof(vkl::Slot);
  if (slots) *slots = vkl::kLedgerSlots;
  if (version) *version = (int32_t)vkl::kLedgerVersion;
}

// what: 0 pid, 1 start time, 2 pid-namespace inode, 3 library instance of the calling process; 4: start time of process `pid`
// (0 if gone), 5: 1 if process `pid` is there (not gone, not a zombie)
uint64_t vk_ledger_self(int32_t what, int64_t pid) {
  if (what >= 4) {
    uint64_t start = 0;
    const int there = vkl::process_state(pid, &start);
    return what == 4 ? start : (uint64_t)there;
  }
  const vkl::Identity me = vkl::self_identity();
  return what == 0 ? (uint64_t)me.pid : what == 1 ? me.start : what == 2 ? me.ns : me.lib;
}

void* vk_ledger_open_at(const char* path, int64_t pid, uint64_t start, uint64_t ns, uint64_t lib, int32_t* status) {
  if (status) *status = vkl::kUnavailable;
  if (!ledger_dev()) return nullptr;
  vkl::Identity me = vkl::self_identity();
  if (pid > 0) {
    me.pid = pid;
    me.start = start;
    me.ns = ns;
  }
  if (lib != 0) me.lib = lib;
  int st = vkl::kUnavailable;
  vkl::Ledger* led = vkl::open_at(path, me, &st);
  if (status) *status = st;
  return led;
}

int32_t vk_ledger_slot(const void* led) { return ledger_dev() && led ? vkl::slot_of(static_cast<const vkl::Ledger*>(led)) : -1; }

int32_t vk_ledger_others(const void* led) { return ledger_dev() && led ? vkl::others(static_cast<const vkl::Ledger*>(led)) : -1; }

uint32_t vk_ledger_generation(const void* led) { return ledger_dev() ? vkl::generation(static_cast<const vkl::Ledger*>(led)) : 0; }

// *process_reserved: what the owner's process holds in all (in / out, as g_poll_reserved is for the library's own contexts)
int32_t vk_ledger_grant(void* led, int32_t* process_reserved, int32_t ctx_reserved, int32_t want) {
  if (!ledger_dev() || !process_reserved) return 0;
  std::atomic<int> process{*process_reserved};
  const int g = vkl::grant(static_cast<vkl::Ledger*>(led), &process, ctx_reserved, want);
  *process_reserved = process.load();
  return g;
}

void vk_ledger_release(void* led, int32_t* process_reserved, int32_t n) {
  if (!ledger_dev() || !process_reserved) return;
  std::atomic<int> process{*process_reserved};
  vkl::release(static_cast<vkl::Ledger*>(led), &process, n);
  *process_reserved = process.load();
}

void vk_ledger_close(void* led, int32_t keep_slot) {
  if (ledger_dev() && led) vkl::close(static_cast<vkl::Ledger*>(led), keep_slot != 0);
}

}  // extern "C"
