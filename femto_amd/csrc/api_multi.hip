// api_multi.hip -- the C ABI's multi-device part: one handle over several GPUs (replicated or striped over their HBM), the
// striped index shared between processes (file descriptors over a Unix socket), and the result gather of the
// one-process-per-GPU form (grouped ncclSend / ncclRecv; RCCL is loaded on first use).  SURVEY.md 8(e); DESIGN.md 6.
#include <dlfcn.h>
#include <fcntl.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "api_internal.hpp"

using namespace femto_amd;

namespace {

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;       // optional: femto_amd_comm_info
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
};

Rccl* rccl() {
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    R.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) R.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!R.lib) return;
    R.GetUniqueId = reinterpret_cast<decltype(R.GetUniqueId)>(dlsym(R.lib, "ncclGetUniqueId"));
    R.CommInitRank = reinterpret_cast<decltype(R.CommInitRank)>(dlsym(R.lib, "ncclCommInitRank"));
    R.CommDestroy = reinterpret_cast<decltype(R.CommDestroy)>(dlsym(R.lib, "ncclCommDestroy"));
    R.Send = reinterpret_cast<decltype(R.Send)>(dlsym(R.lib, "ncclSend"));
    R.Recv = reinterpret_cast<decltype(R.Recv)>(dlsym(R.lib, "ncclRecv"));
    R.GroupStart = reinterpret_cast<decltype(R.GroupStart)>(dlsym(R.lib, "ncclGroupStart"));
    R.GroupEnd = reinterpret_cast<decltype(R.GroupEnd)>(dlsym(R.lib, "ncclGroupEnd"));
    R.GetErrorString = reinterpret_cast<decltype(R.GetErrorString)>(dlsym(R.lib, "ncclGetErrorString"));
    R.CommCount = reinterpret_cast<decltype(R.CommCount)>(dlsym(R.lib, "ncclCommCount"));
    R.CommUserRank = reinterpret_cast<decltype(R.CommUserRank)>(dlsym(R.lib, "ncclCommUserRank"));
  });
  if (!R.lib || !R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.Send || !R.Recv || !R.GroupStart || !R.GroupEnd) return nullptr;
  return &R;
}

#define RCCL_TRY(R, expr)                                                                                         \
  do {                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                     \
    if (r_ != ncclSuccess)                                                                                        \
      return set_err(FEMTO_AMD_ERR_INVALID, std::string(#expr) + ": " + ((R)->GetErrorString ? (R)->GetErrorString(r_) : "rccl error")); \
  } while (0)

}  // namespace

namespace femto_amd {
void comm_destroy(femto_amd_index* ix) {       // femto_amd_close
  if (!ix->comm) return;
  if (Rccl* R = rccl()) (void)R->CommDestroy(ix->comm);
  ix->comm = nullptr;
}
}  // namespace femto_amd

extern "C" {

// ---- several GPUs ------------------------------------------------------------------------------------------------------
int femto_amd_open_multi(const char* index_path, int ndev, const int* devices, femto_amd_index_t** out) {
  API_BEGIN
  if (!index_path || !out || ndev < 1 || ndev > 64 || !devices) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  *out = nullptr;
  femto_amd_index_t* ix = nullptr;
  int rc = femto_amd_open(index_path, -1, &ix);     // parse-only: facts, document table
  if (rc) return rc;
  ix->children.assign(size_t(ndev), nullptr);
  std::vector<int> rcs(size_t(ndev), 0);
  std::vector<std::string> msgs((size_t(ndev)));
  std::vector<std::thread> th;
  for (int i = 0; i < ndev; i++)
    th.emplace_back([&, i] {
      rcs[size_t(i)] = femto_amd_open(index_path, devices[i], &ix->children[size_t(i)]);
      if (rcs[size_t(i)]) msgs[size_t(i)] = femto_amd_last_error();
    });
  for (auto& t : th) t.join();
  for (int i = 0; i < ndev; i++)
    if (rcs[size_t(i)]) {
      const int code = rcs[size_t(i)];
      const std::string m = "device " + std::to_string(devices[i]) + ": " + msgs[size_t(i)];
      for (auto& c : ix->children) if (!c) c = nullptr;
      std::vector<femto_amd_index*> kids;
      for (femto_amd_index* c : ix->children) if (c) kids.push_back(c);
      ix->children = kids;
      femto_amd_close(ix);
      return set_err(code, m);
    }
  *out = ix;
  return FEMTO_AMD_OK;
  API_END
}

// A view of `b` (the builder of a striped index) for another GPU: the big arrays are the builder's own address ranges
// (their pages are mapped for every listed GPU), the small tables are copied into the view's GPU.
// the small-table pointers of a view's DevIndex (copied from the builder's) -> the view's own copies
static void remap_small_tables(femto_amd_index* v, const std::vector<std::pair<const void*, void*>>& map) {
  auto remap = [&](auto*& ptr) {
    for (auto& m : map)
      if (m.first == static_cast<const void*>(ptr)) { ptr = static_cast<std::remove_reference_t<decltype(ptr)>>(m.second); return; }
  };
  DevIndex& d = v->dev;
  remap(d.nodes); remap(d.buckets); remap(d.seqs); remap(d.occ_base); remap(d.leaf_code); remap(d.C); remap(d.cum); remap(d.hint);
  remap(d.bdir); remap(d.lnodes); remap(d.lseqs); remap(d.occ); remap(d.pack_code); remap(d.pack_c); remap(d.p2_base); remap(d.p2_c);
  remap(d.p2_code); remap(d.p2_alpha);
  remap(v->d_dense);
}

static int make_view(femto_amd_index* b, int device, femto_amd_index** out) {
  femto_amd_index* v = new (std::nothrow) femto_amd_index();
  if (!v) return set_err(FEMTO_AMD_ERR_MEM, "out of memory");
  auto fail = [&](int code) { femto_amd_close(v); return code; };
  v->borrowed = true;
  v->device = device;
  HostIndex& h = v->host;
  const HostIndex& s = b->host;
  h.total_length = s.total_length; h.number_of_blocks = s.number_of_blocks; h.number_of_documents = s.number_of_documents;
  h.block_size = s.block_size; h.b_size = s.b_size; h.mark_period = s.mark_period; h.chunk_size = s.chunk_size;
  h.text_size_bits = s.text_size_bits; h.buckets_per_block = s.buckets_per_block; h.total_buckets = s.total_buckets;
  h.header = s.header; h.C = s.C; h.doc_ends = s.doc_ends; h.doc_info_off = s.doc_info_off; h.dir_regular = s.dir_regular;
  h.block_off = s.block_off; h.block_len = s.block_len;
  v->opt = b->opt;
  v->mode = b->mode; v->sort_queries = b->sort_queries; v->dense_bits = b->dense_bits;
  v->dense_sigma = b->dense_sigma; v->sort_min = b->sort_min; v->h_dense = b->h_dense; v->table_bytes = b->table_bytes;
  v->ktab2_bytes = b->ktab2_bytes; v->ctx_bytes = b->ctx_bytes; v->ctx2_bytes = b->ctx2_bytes; v->ctxm_bytes = b->ctxm_bytes; v->n_marks = b->n_marks; v->p2_lines1 = b->p2_lines1; v->p2_lines2 = b->p2_lines2;
  v->ind_bytes = b->ind_bytes; v->text_bytes = b->text_bytes; v->pack_bytes = b->pack_bytes; v->pack2_bytes = b->pack2_bytes;
  if (hipSetDevice(device) != hipSuccess) return fail(set_err(FEMTO_AMD_ERR_INVALID, "no usable HIP device " + std::to_string(device)));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) v->num_cus = prop.multiProcessorCount;
  std::vector<std::pair<const void*, void*>> map;
  for (auto& t : b->small_tables) {
    void* q = nullptr;
    if (hipMalloc(&q, t.second) != hipSuccess) return fail(set_err(FEMTO_AMD_ERR_MEM, "hipMalloc (small table copy)"));
    v->owned_small.push_back(q);
    if (hipMemcpyPeer(q, device, t.first, b->device, t.second) != hipSuccess)
      return fail(set_err(FEMTO_AMD_ERR_INVALID, "hipMemcpyPeer (small table copy)"));
    map.emplace_back(t.first, q);
  }
  v->dev = b->dev;
  v->d_dense = b->d_dense;
  remap_small_tables(v, map);
  *out = v;
  return 0;
}

int femto_amd_open_multi_striped(const char* index_path, int ndev, const int* devices, femto_amd_index_t** out) {
  API_BEGIN
  if (!index_path || !out || ndev < 1 || ndev > 64 || !devices) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  *out = nullptr;
  femto_amd_index_t* ix = nullptr;
  int rc = femto_amd_open(index_path, -1, &ix);     // parse-only: facts, document table
  if (rc) return rc;
  const std::vector<int> devs(devices, devices + ndev);
  for (int a : devs)      // every GPU reads the others' stripes directly
    for (int b : devs)
      if (a != b) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
          femto_amd_close(ix);
          return set_err(FEMTO_AMD_ERR_INVALID, "no peer access between devices " + std::to_string(a) + " and " + std::to_string(b));
        }
      }
  femto_amd_index* builder = nullptr;
  rc = open_impl(index_path, devs[0], 0, 0, &builder, &devs);     // derives everything on the first GPU, into striped arrays
  if (rc) { femto_amd_close(ix); return rc; }
  ix->children.push_back(builder);
  for (int i = 1; i < ndev; i++) {
    femto_amd_index* v = nullptr;
    if ((rc = make_view(builder, devs[size_t(i)], &v))) { femto_amd_close(ix); return rc; }
    ix->children.push_back(v);
  }
  *out = ix;
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_device_count(const femto_amd_index_t* ix) {
  if (!ix) return 0;
  return ix->children.empty() ? (ix->device >= 0 ? 1 : 0) : int(ix->children.size());
}

// ---- a striped index shared between PROCESSES (one process per GPU) --------------------------------------------------
// The process that built the striped index (femto_amd_open_multi_striped) exports every physical stripe as a POSIX file
// descriptor (hipMemExportToShareableHandle) and hands them, with a description of the handle, to the other processes over
// a Unix-domain socket (SCM_RIGHTS); each of those imports the stripes, maps them AT THE SAME ADDRESSES (so the builder's
// DevIndex is valid as it stands), copies the small tables to its own GPU and is then an ordinary single-GPU handle whose
// big arrays live in the HBM of all the GPUs: same kernels, same fast paths, remote lines over xGMI.
extern "C++" {
namespace {

struct SharedFacts {       // what make_view copies from the builder, as plain data
  int32_t mode, sort_queries, dense_bits;
  double dense_sigma;
  int64_t sort_min, table_bytes, ktab2_bytes, ctx_bytes, ctx2_bytes, ctxm_bytes, n_marks, p2_lines1, p2_lines2, ind_bytes, text_bytes, pack_bytes, pack2_bytes;
  uint64_t d_dense;
  uint64_t dev_bytes, n_dense, n_ranges, n_small, n_fds;
};
// ("femtoSHR" + the version of the derived layouts a description describes: a client of another build -- other dense codes, other
// line formats behind the same DevIndex size -- is refused instead of searching with the wrong tables.  Bump with every change of
// a derived layout or of the key fields; round 6: 2 -- frequency-ordered dense codes of byte alphabets, one-row level-table entries)
constexpr uint64_t kDerivedLayoutVersion = 2;
constexpr uint64_t kSharedMagic = 0x66656d746f534852ull + kDerivedLayoutVersion;

void put(std::string& b, const void* p, size_t n) { b.append(static_cast<const char*>(p), n); }
template <class T> void put(std::string& b, const T& v) { put(b, &v, sizeof v); }

struct Reader {
  const char* p;
  size_t left;
  bool get(void* out, size_t n) {
    if (n > left) return false;
    memcpy(out, p, n);
    p += n;
    left -= n;
    return true;
  }
  template <class T> bool get(T& v) { return get(&v, sizeof v); }
};

bool send_all(int fd, const void* buf, size_t n) {
  const char* c = static_cast<const char*>(buf);
  while (n) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k;
    n -= size_t(k);
  }
  return true;
}
bool recv_all(int fd, void* buf, size_t n) {
  char* c = static_cast<char*>(buf);
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
    c += k;
    n -= size_t(k);
  }
  return true;
}
// hipMemImportFromShareableHandle takes the file descriptor BY VALUE in the pointer argument, as CUDA does, from HIP 7.2 on;
// HIP 7.0 (the runtime PyTorch 2.10+rocm7.0 brings along, which is the one in the process when the library is used from
// Python) dereferences the argument as an int* and crashes on a value.  Both verified on the MI355X box (tools/vmm_probe.cpp).
hipError_t import_shareable(hipMemGenericAllocationHandle_t* h, int fd) {
  int ver = 0;
  if (hipRuntimeGetVersion(&ver) != hipSuccess) ver = 0;
  if (ver < 70200000) return hipMemImportFromShareableHandle(h, &fd, hipMemHandleTypePosixFileDescriptor);
  return hipMemImportFromShareableHandle(h, reinterpret_cast<void*>(static_cast<uintptr_t>(fd)), hipMemHandleTypePosixFileDescriptor);
}

constexpr int kFdsPerMsg = 32;
bool send_fds(int sock, const int* fds, int n) {
  for (int at = 0; at < n; at += kFdsPerMsg) {
    const int k = std::min(kFdsPerMsg, n - at);
    struct msghdr msg;
    memset(&msg, 0, sizeof msg);
    char tag = 'F';
    struct iovec io = {&tag, 1};
    alignas(struct cmsghdr) char buf[CMSG_SPACE(sizeof(int) * kFdsPerMsg)];
    memset(buf, 0, sizeof buf);
    msg.msg_iov = &io;
    msg.msg_iovlen = 1;
    msg.msg_control = buf;
    msg.msg_controllen = CMSG_SPACE(sizeof(int) * size_t(k));
    struct cmsghdr* c = CMSG_FIRSTHDR(&msg);
    c->cmsg_level = SOL_SOCKET;
    c->cmsg_type = SCM_RIGHTS;
    c->cmsg_len = CMSG_LEN(sizeof(int) * size_t(k));
    memcpy(CMSG_DATA(c), fds + at, sizeof(int) * size_t(k));
    if (::sendmsg(sock, &msg, MSG_NOSIGNAL) < 0) return false;
  }
  return true;
}
bool recv_fds(int sock, int* fds, int n) {
  for (int at = 0; at < n; at += kFdsPerMsg) {
    const int k = std::min(kFdsPerMsg, n - at);
    struct msghdr msg;
    memset(&msg, 0, sizeof msg);
    char tag = 0;
    struct iovec io = {&tag, 1};
    alignas(struct cmsghdr) char buf[CMSG_SPACE(sizeof(int) * kFdsPerMsg)];
    memset(buf, 0, sizeof buf);
    msg.msg_iov = &io;
    msg.msg_iovlen = 1;
    msg.msg_control = buf;
    msg.msg_controllen = sizeof buf;
    if (::recvmsg(sock, &msg, 0) <= 0) return false;
    struct cmsghdr* c = CMSG_FIRSTHDR(&msg);
    if (!c || c->cmsg_level != SOL_SOCKET || c->cmsg_type != SCM_RIGHTS || c->cmsg_len != CMSG_LEN(sizeof(int) * size_t(k))) return false;
    memcpy(fds + at, CMSG_DATA(c), sizeof(int) * size_t(k));
  }
  return true;
}

}  // namespace
}  // extern "C++"

int femto_amd_striped_serve(femto_amd_index_t* ix, const char* socket_path, int nclients) {
  API_BEGIN
  if (!ix || !socket_path || nclients < 0) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (ix->children.empty() || ix->children[0]->striped.empty()) return set_err(FEMTO_AMD_ERR_INVALID, "not a striped handle (femto_amd_open_multi_striped)");
  femto_amd_index* b = ix->children[0];
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  SharedFacts f{};
  f.mode = b->mode; f.sort_queries = b->sort_queries; f.dense_bits = b->dense_bits;
  f.dense_sigma = b->dense_sigma; f.sort_min = b->sort_min;
  f.table_bytes = b->table_bytes; f.ktab2_bytes = b->ktab2_bytes; f.ctx_bytes = b->ctx_bytes; f.ctx2_bytes = b->ctx2_bytes; f.ctxm_bytes = b->ctxm_bytes;
  f.n_marks = b->n_marks; f.p2_lines1 = b->p2_lines1; f.p2_lines2 = b->p2_lines2; f.ind_bytes = b->ind_bytes;
  f.text_bytes = b->text_bytes; f.pack_bytes = b->pack_bytes; f.pack2_bytes = b->pack2_bytes;
  f.d_dense = reinterpret_cast<uint64_t>(b->d_dense);
  f.dev_bytes = sizeof(DevIndex); f.n_dense = b->h_dense.size(); f.n_ranges = b->striped.size(); f.n_small = b->small_tables.size();
  std::vector<int> fds;
  auto close_fds = [&]() { for (int fd : fds) ::close(fd); };
  for (auto& st : b->striped)
    for (auto& h : st.handles) {
      int fd = -1;
      const hipError_t e = hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0);
      if (e != hipSuccess) { close_fds(); return set_err(FEMTO_AMD_ERR_INVALID, std::string("hipMemExportToShareableHandle: ") + hipGetErrorString(e)); }
      fds.push_back(fd);
    }
  f.n_fds = fds.size();
  std::string blob;
  put(blob, kSharedMagic);
  put(blob, f);
  put(blob, &b->dev, sizeof(DevIndex));
  if (!b->h_dense.empty()) put(blob, b->h_dense.data(), b->h_dense.size());
  for (auto& st : b->striped) {
    const uint64_t r[4] = {reinterpret_cast<uint64_t>(st.va), uint64_t(st.size), uint64_t(st.chunk), uint64_t(st.handles.size())};
    put(blob, r, sizeof r);
  }
  std::vector<char> tmp;
  for (auto& t : b->small_tables) {
    const uint64_t r[2] = {reinterpret_cast<uint64_t>(t.first), uint64_t(t.second)};
    put(blob, r, sizeof r);
    tmp.resize(t.second);
    if (hipMemcpy(tmp.data(), t.first, t.second, hipMemcpyDeviceToHost) != hipSuccess) { close_fds(); return set_err(FEMTO_AMD_ERR_INVALID, "hipMemcpy (small table)"); }
    put(blob, tmp.data(), t.second);
  }
  struct sockaddr_un addr;
  memset(&addr, 0, sizeof addr);
  addr.sun_family = AF_UNIX;
  if (strlen(socket_path) >= sizeof addr.sun_path) { close_fds(); return set_err(FEMTO_AMD_ERR_PARAM, "socket path too long"); }
  strcpy(addr.sun_path, socket_path);
  const int ls = ::socket(AF_UNIX, SOCK_STREAM, 0);
  if (ls < 0) { close_fds(); return set_err(FEMTO_AMD_ERR_IO, "socket()"); }
  // The descriptors handed out give read-write access to this process's GPU memory: the socket file is created 0600 --
  // fchmod() on the unbound socket sets the mode bind() creates the file with (Linux: the socket inode's mode, less the
  // umask), so the process-wide umask is never touched (other threads of a Python / PyTorch host create files meanwhile) --
  // and every client must run as this user (SO_PEERCRED).  A stale socket file of an earlier run is replaced only when
  // bind() says the address is in use, and only if the path still IS a socket then.  A client rank that died before
  // attaching must not hang the builder: accept() and the hand-shake wait at most FEMTO_AMD_STRIPED_TIMEOUT seconds (600).
  (void)::fchmod(ls, 0600);
  int bind_rc = ::bind(ls, reinterpret_cast<struct sockaddr*>(&addr), sizeof addr);
  if (bind_rc && errno == EADDRINUSE) {
    struct stat sb;
    if (::lstat(socket_path, &sb) != 0 || !S_ISSOCK(sb.st_mode)) {
      ::close(ls);
      close_fds();
      return set_err(FEMTO_AMD_ERR_PARAM, std::string(socket_path) + " exists and is not a socket");
    }
    ::unlink(socket_path);
    bind_rc = ::bind(ls, reinterpret_cast<struct sockaddr*>(&addr), sizeof addr);
  }
  if (bind_rc == 0) (void)::chmod(socket_path, 0600);      // (belt and braces on kernels that ignore the fchmod)
  if (bind_rc || ::listen(ls, 64)) {
    ::close(ls);
    close_fds();
    return set_err(FEMTO_AMD_ERR_IO, std::string("bind/listen ") + socket_path + ": " + strerror(errno));
  }
  int timeout_s = 600;
  if (const char* e = getenv("FEMTO_AMD_STRIPED_TIMEOUT")) timeout_s = std::max(1, atoi(e));
  int rc = FEMTO_AMD_OK;
  for (int c = 0; c < nclients && rc == FEMTO_AMD_OK; c++) {
    struct pollfd pfd{ls, POLLIN, 0};
    const int pr = ::poll(&pfd, 1, timeout_s * 1000);
    if (pr <= 0) { rc = set_err(FEMTO_AMD_ERR_IO, pr == 0 ? "timed out waiting for a client to attach the striped index" : "poll()"); break; }
    const int cs = ::accept(ls, nullptr, nullptr);
    if (cs < 0) { rc = set_err(FEMTO_AMD_ERR_IO, "accept()"); break; }
    struct timeval tv{timeout_s, 0};
    (void)::setsockopt(cs, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
    (void)::setsockopt(cs, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof tv);
    struct ucred cred;
    socklen_t clen = sizeof cred;
    if (::getsockopt(cs, SOL_SOCKET, SO_PEERCRED, &cred, &clen) != 0 || cred.uid != ::geteuid()) {
      ::close(cs);
      rc = set_err(FEMTO_AMD_ERR_INVALID, "a process of another user tried to attach the striped index");
      break;
    }
    const uint64_t len = blob.size();
    char ack = 0;
    if (!send_all(cs, &len, sizeof len) || !send_all(cs, blob.data(), blob.size()) || !send_fds(cs, fds.data(), int(fds.size())) ||
        !recv_all(cs, &ack, 1) || ack != 'K')
      rc = set_err(FEMTO_AMD_ERR_IO, "a client did not attach the striped index");
    ::close(cs);
  }
  ::close(ls);
  ::unlink(socket_path);
  close_fds();
  return rc;
  API_END
}

int femto_amd_open_striped_client(const char* index_path, const char* socket_path, int device, int timeout_s, femto_amd_index_t** out) {
  API_BEGIN
  if (!index_path || !socket_path || !out || device < 0) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  *out = nullptr;
  femto_amd_index* v = nullptr;
  int rc = open_impl(index_path, -1, 0, 0, &v);      // parse-only: the index's facts and document table
  if (rc) return rc;
  std::vector<int> fds;
  auto fail = [&](int code) {
    for (int fd : fds) if (fd >= 0) ::close(fd);
    femto_amd_close(v);
    return code;
  };
  v->host.image.clear();
  v->host.image.shrink_to_fit();
  v->borrowed = true;
  v->imported = true;
  v->device = device;
  if (hipSetDevice(device) != hipSuccess) return fail(set_err(FEMTO_AMD_ERR_INVALID, "no usable HIP device " + std::to_string(device)));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) v->num_cus = prop.multiProcessorCount;
  struct sockaddr_un addr;
  memset(&addr, 0, sizeof addr);
  addr.sun_family = AF_UNIX;
  if (strlen(socket_path) >= sizeof addr.sun_path) return fail(set_err(FEMTO_AMD_ERR_PARAM, "socket path too long"));
  strcpy(addr.sun_path, socket_path);
  int cs = -1;
  for (int tries = 0; tries < std::max(1, timeout_s) * 10; tries++) {    // the builder may still be deriving the index
    cs = ::socket(AF_UNIX, SOCK_STREAM, 0);
    if (cs < 0) return fail(set_err(FEMTO_AMD_ERR_IO, "socket()"));
    if (::connect(cs, reinterpret_cast<struct sockaddr*>(&addr), sizeof addr) == 0) break;
    ::close(cs);
    cs = -1;
    std::this_thread::sleep_for(std::chrono::milliseconds(100));
  }
  if (cs < 0) return fail(set_err(FEMTO_AMD_ERR_IO, std::string("no builder at ") + socket_path));
  uint64_t len = 0;
  std::string blob;
  if (!recv_all(cs, &len, sizeof len) || len < 16 || len > (uint64_t(1) << 32)) { ::close(cs); return fail(set_err(FEMTO_AMD_ERR_IO, "bad description")); }
  blob.resize(len);
  if (!recv_all(cs, &blob[0], len)) { ::close(cs); return fail(set_err(FEMTO_AMD_ERR_IO, "short description")); }
  Reader R{blob.data(), blob.size()};
  uint64_t magic = 0;
  SharedFacts f{};
  if (!R.get(magic) || magic != kSharedMagic || !R.get(f) || f.dev_bytes != sizeof(DevIndex) || f.n_fds > 4096 || f.n_ranges > 256 || f.n_small > 4096) {
    ::close(cs);
    return fail(set_err(FEMTO_AMD_ERR_FORMAT, "description of another library version"));
  }
  fds.assign(size_t(f.n_fds), -1);
  if (!recv_fds(cs, fds.data(), int(fds.size()))) { ::close(cs); return fail(set_err(FEMTO_AMD_ERR_IO, "file descriptors did not arrive")); }
  auto bail = [&](int code) { ::close(cs); return fail(code); };
  if (!R.get(&v->dev, sizeof(DevIndex))) return bail(set_err(FEMTO_AMD_ERR_FORMAT, "short description"));
  v->h_dense.resize(size_t(f.n_dense));
  if (f.n_dense && !R.get(v->h_dense.data(), size_t(f.n_dense))) return bail(set_err(FEMTO_AMD_ERR_FORMAT, "short description"));
  v->mode = f.mode; v->sort_queries = f.sort_queries != 0; v->dense_bits = f.dense_bits;
  v->dense_sigma = f.dense_sigma; v->sort_min = f.sort_min;
  v->table_bytes = f.table_bytes; v->ktab2_bytes = f.ktab2_bytes; v->ctx_bytes = f.ctx_bytes; v->ctx2_bytes = f.ctx2_bytes; v->ctxm_bytes = f.ctxm_bytes;
  v->n_marks = f.n_marks; v->p2_lines1 = f.p2_lines1; v->p2_lines2 = f.p2_lines2; v->ind_bytes = f.ind_bytes;
  v->text_bytes = f.text_bytes; v->pack_bytes = f.pack_bytes; v->pack2_bytes = f.pack2_bytes;
  v->d_dense = reinterpret_cast<uint8_t*>(f.d_dense);
  size_t next_fd = 0;
  for (uint64_t k = 0; k < f.n_ranges; k++) {
    uint64_t r[4];
    if (!R.get(r, sizeof r) || r[3] == 0 || r[3] > 64 || r[1] != r[2] * r[3] || next_fd + r[3] > fds.size())
      return bail(set_err(FEMTO_AMD_ERR_FORMAT, "bad stripe table"));
    femto_amd_index::Striped st;
    st.va = nullptr;
    st.size = size_t(r[1]);
    st.chunk = size_t(r[2]);
    void* want = reinterpret_cast<void*>(r[0]);
    hipError_t e = hipMemAddressReserve(&st.va, st.size, 0, want, 0);
    if (e != hipSuccess || st.va != want) {
      if (e == hipSuccess) (void)hipMemAddressFree(st.va, st.size);
      return bail(set_err(FEMTO_AMD_ERR_INVALID, "the builder's address range is not free in this process"));
    }
    for (uint64_t i = 0; i < r[3] && e == hipSuccess; i++) {
      hipMemGenericAllocationHandle_t h;
      e = import_shareable(&h, fds[next_fd + i]);
      if (e != hipSuccess) break;
      st.handles.push_back(h);
      e = hipMemMap(static_cast<char*>(st.va) + size_t(i) * st.chunk, st.chunk, 0, h, 0);
    }
    next_fd += r[3];
    if (e == hipSuccess) {
      hipMemAccessDesc a{};
      a.location.type = hipMemLocationTypeDevice;
      a.location.id = device;
      a.flags = hipMemAccessFlagsProtReadWrite;
      e = hipMemSetAccess(st.va, st.size, &a, 1);
    }
    v->striped.push_back(st);      // (released by femto_amd_close, mapped or not)
    if (e != hipSuccess) return bail(set_err(FEMTO_AMD_ERR_INVALID, std::string("mapping a shared stripe: ") + hipGetErrorString(e)));
  }
  for (int& fd : fds) { ::close(fd); fd = -1; }      // the imported handles hold the memory now
  std::vector<std::pair<const void*, void*>> map;
  std::vector<char> tmp;
  for (uint64_t k = 0; k < f.n_small; k++) {
    uint64_t r[2];
    if (!R.get(r, sizeof r) || r[1] > R.left) return bail(set_err(FEMTO_AMD_ERR_FORMAT, "bad small-table list"));
    void* q = nullptr;
    if (hipMalloc(&q, size_t(r[1]) ? size_t(r[1]) : 16) != hipSuccess) return bail(set_err(FEMTO_AMD_ERR_MEM, "hipMalloc (small table copy)"));
    v->owned_small.push_back(q);
    if (r[1] && hipMemcpy(q, R.p, size_t(r[1]), hipMemcpyHostToDevice) != hipSuccess) return bail(set_err(FEMTO_AMD_ERR_INVALID, "hipMemcpy (small table copy)"));
    R.p += r[1];
    R.left -= size_t(r[1]);
    map.emplace_back(reinterpret_cast<const void*>(r[0]), q);
  }
  remap_small_tables(v, map);
  if (const hipError_t se = hipDeviceSynchronize(); se != hipSuccess)
    return bail(set_err(FEMTO_AMD_ERR_INVALID, std::string("hipDeviceSynchronize: ") + hipGetErrorString(se)));
  const char ack = 'K';
  (void)send_all(cs, &ack, 1);
  ::close(cs);
  *out = v;
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_multi_child(femto_amd_index_t* ix, int i, femto_amd_index_t** child) {
  if (!ix || !child || i < 0 || size_t(i) >= ix->children.size()) return set_err(FEMTO_AMD_ERR_PARAM, "no such replica");
  *child = ix->children[size_t(i)];
  return FEMTO_AMD_OK;
}

int femto_amd_comm_unique_id(void* id128) {
  API_BEGIN
  if (!id128) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  Rccl* R = rccl();
  if (!R) return set_err(FEMTO_AMD_ERR_MISSING, "librccl.so could not be loaded");
  static_assert(sizeof(ncclUniqueId) == 128, "id blob layout");
  ncclUniqueId id;
  RCCL_TRY(R, R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return FEMTO_AMD_OK;
  API_END
}

int femto_amd_comm_init(femto_amd_index_t* ix, const void* id128, int nranks, int rank) {
  API_BEGIN
  if (!ix || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  int rc = ensure_device(ix);
  if (rc) return rc;
  Rccl* R = rccl();
  if (!R) return set_err(FEMTO_AMD_ERR_MISSING, "librccl.so could not be loaded");
  if (ix->comm) return set_err(FEMTO_AMD_ERR_INVALID, "communicator already initialised");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  RCCL_TRY(R, R->CommInitRank(&ix->comm, nranks, id, rank));
  ix->comm_rank = rank;
  ix->comm_size = nranks;
  return FEMTO_AMD_OK;
  API_END
}

// what the communicator itself says (ncclCommCount / ncclCommUserRank), next to what femto_amd_comm_init was told
int femto_amd_comm_info(femto_amd_index_t* ix, int* nranks, int* rank) {
  API_BEGIN
  if (!ix || !nranks || !rank) return set_err(FEMTO_AMD_ERR_PARAM, "null argument");
  if (!ix->comm) return set_err(FEMTO_AMD_ERR_INVALID, "no communicator: call femto_amd_comm_init first");
  Rccl* R = rccl();
  *nranks = ix->comm_size;
  *rank = ix->comm_rank;
  if (R && R->CommCount && R->CommUserRank) {
    RCCL_TRY(R, R->CommCount(ix->comm, nranks));
    RCCL_TRY(R, R->CommUserRank(ix->comm, rank));
  }
  return FEMTO_AMD_OK;
  API_END
}

// bytes_per_rank bytes of every rank land, in rank order, in d_recv on `root`: one grouped batch of point-to-point
// transfers (xGMI is point to point: every peer's payload crosses its own link into the root)
int femto_amd_comm_gather(femto_amd_index_t* ix, const void* d_send, void* d_recv, int64_t bytes_per_rank, int root, void* stream_) {
  API_BEGIN
  if (!ix || !ix->comm) return set_err(FEMTO_AMD_ERR_INVALID, "no communicator (femto_amd_comm_init)");
  if (bytes_per_rank < 0 || root < 0 || root >= ix->comm_size || (bytes_per_rank && !d_send)) return set_err(FEMTO_AMD_ERR_PARAM, "bad arguments");
  if (ix->comm_rank == root && bytes_per_rank && !d_recv) return set_err(FEMTO_AMD_ERR_PARAM, "root needs a receive buffer");
  int rc = ensure_device(ix);
  if (rc) return rc;
  Rccl* R = rccl();
  if (!R) return set_err(FEMTO_AMD_ERR_MISSING, "librccl.so could not be loaded");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (bytes_per_rank == 0) return FEMTO_AMD_OK;
  RCCL_TRY(R, R->GroupStart());
  ncclResult_t r1 = ncclSuccess;
  if (ix->comm_rank == root) {
    for (int p = 0; p < ix->comm_size && r1 == ncclSuccess; p++) {
      char* dst = static_cast<char*>(d_recv) + size_t(p) * size_t(bytes_per_rank);
      if (p == root) {
        if (hipMemcpyAsync(dst, d_send, size_t(bytes_per_rank), hipMemcpyDeviceToDevice, stream) != hipSuccess) r1 = ncclUnhandledCudaError;
      } else {
        r1 = R->Recv(dst, size_t(bytes_per_rank), ncclInt8, p, ix->comm, stream);
      }
    }
  } else {
    r1 = R->Send(d_send, size_t(bytes_per_rank), ncclInt8, root, ix->comm, stream);
  }
  ncclResult_t r2 = R->GroupEnd();
  RCCL_TRY(R, r1);
  RCCL_TRY(R, r2);
  return FEMTO_AMD_OK;
  API_END
}

}  // extern "C"
