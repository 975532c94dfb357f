// vk_ledger.h - the bound behind the polling hand-off, host side (no HIP, no device code): the launch rule and the grant as pure
// functions (exported in the ABI: vk_poll_rule / vk_poll_grant / vk_poll_budget), and the device-wide ledger of reserved waiters
// that the processes of one user on one host share in /dev/shm (vk_ledger.cpp; include/victor_hip.h).
#pragma once

#include <atomic>
#include <cstdint>
#include <string>

namespace vkl __attribute__((visibility("hidden"))) {       // (internal: not part of the library's exported symbols)

// Points per launch whose split work is handed over by polling (TheoryArgs::poll): one waiting workgroup per point.  Few, so
// that the waiting workgroups of every launch in flight on the GPU - other contexts, other processes - can never fill an
// XCD (64 workgroup slots at least) and keep the workgroups they wait for off it.
constexpr long long kPollPoints = 8;
// The bound behind "never fill an XCD" (include/victor_hip.h: vk_poll_rule).  A polling launch is taken only
// when at least two of its workgroups fit on a CU (LDS and launch bounds), so an XCD of 32 CUs has kPollXcdSlots = 64 workgroup
// slots at least; a deadlock needs one XCD's slots ALL held by waiting workgroups (one per point of a polling launch in flight)
// whose producers cannot be placed, so fewer than 64 waiters resident on the whole device exclude it.  What the library
// enforces is its own process's share: every context reserves the waiters its polling launches may have resident (its launches
// are stream-ordered: never more than one in flight) out of kPollBudget = 32 per process - an owner process's default four
// contexts x eight requests -; a launch whose context holds no reservation for its points hands over through the completion
// counters instead (the same sums in the same order: not a bit changes).  Across processes the sum of the reservations must
// stay below 64 - ONE process with the full budget (the GPU owner of section 6) plus up to 31 single-point contexts of other
// processes, or up to 63 processes that each evaluate one point per call in one context -, kept in the ledger below; a
// process beyond the bound simply gets no reservation.
constexpr int kPollXcdSlots = 64;
constexpr int kPollBudget = 32;

// ---- who owns a slot ------------------------------------------------------------------------------------------------------
// A pid alone does not name a process: pids are recycled, and containers that share /dev/shm but not the pid namespace see each
// other's numbers as dead (or as somebody else).  A slot's owner is therefore {pid, start time of that pid (field 22 of
// /proc/<pid>/stat, clock ticks since boot), inode of its pid namespace (/proc/self/ns/pid), library instance}: the last
// because two copies of this library in one process (the product and its development twin, tests/devlib.py) each keep their
// own count of reserved waiters and must not overwrite each other's.
struct Identity {
  int64_t pid = 0;
  uint64_t start = 0;     // 0: unknown (no /proc)
  uint64_t ns = 0;        // 0: unknown
  uint64_t lib = 0;
};
Identity self_identity();

// state of process `pid` as this process sees it: 0 = gone (ESRCH) or a zombie, 1 = there; *start = its start time or 0
int process_state(int64_t pid, uint64_t* start);

enum OpenStatus {
  kOpened = 0,
  kUnavailable = 1,   // the file cannot be created or mapped (no /dev/shm, read-only): the process budget alone applies
  kUntrusted = 2,     // the file is there but is not ours to trust (another owner, another mode, a symbolic link, foreign contents,
                      // a lock nobody releases): NO polling for this process - what the device holds cannot be known
  kFull = 3,          // every slot belongs to a living (or foreign) owner: no polling for this process
};

struct Ledger;        // one mapped ledger file and this owner's slot in it

// Open (create) the ledger at `path` and claim a slot for `me`: an own slot left by this very owner is adopted as it is
// (its reservations stand), a free slot is taken, a slot whose owner is gone - dead, a zombie, or the pid now names a process
// with another start time - is taken over.  Slots of another pid namespace are never taken over.
Ledger* open_at(const char* path, const Identity& me, int* status);
// keep_slot: leave the slot behind as a killed process would (tests); otherwise the slot is freed
void close(Ledger* led, bool keep_slot);
int slot_of(const Ledger* led);
// waiters reserved by every OTHER owner that is living or cannot be judged (another pid namespace)
int others(const Ledger* led);
// bumped whenever reservations are returned or a slot changes hands: a context that was refused asks again only after it moved
uint32_t generation(const Ledger* led);

// The grant of `want` waiters to a context that holds `ctx_reserved`, all or nothing (vk_poll_grant), counted in the process-wide
// `process` and - with a ledger - published device-wide, re-checked against what the other owners published meanwhile and taken
// back if two raced past the bound (both may: conservative).  `led` may be NULL (no ledger: the process budget alone).
int grant(Ledger* led, std::atomic<int>* process, int ctx_reserved, int want);
void release(Ledger* led, std::atomic<int>* process, int n);

// the ledger of the GPU with PCI bus id `bus` for this process (opened once; thread-safe); NULL with *status telling why
Ledger* for_device(const std::string& bus, int* status);

}  // namespace vkl
