// tools/vmm_probe.cpp -- can a HIP virtual-memory allocation be shared between two PROCESSES on this box?
// parent: hipMemCreate + map + fill, hipMemExportToShareableHandle (POSIX fd), fd sent over a socketpair (SCM_RIGHTS)
// child : hipMemImportFromShareableHandle, hipMemAddressReserve AT THE PARENT'S ADDRESS, map, read back.
// Build: hipcc -O2 --offload-arch=gfx950 -o vmm_probe tools/vmm_probe.cpp ; run: ./vmm_probe
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s -> %s\n", role, #x, hipGetErrorString(e_)); return 2; } } while (0)
static const char* role = "parent";
static int send_fd(int sock, int fd, uint64_t addr, uint64_t size) {
  struct msghdr msg; std::memset(&msg, 0, sizeof msg);
  uint64_t payload[2] = {addr, size};
  struct iovec io = {payload, sizeof payload};
  char buf[CMSG_SPACE(sizeof(int))]; std::memset(buf, 0, sizeof buf);
  msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = buf; msg.msg_controllen = sizeof buf;
  struct cmsghdr* c = CMSG_FIRSTHDR(&msg);
  c->cmsg_level = SOL_SOCKET; c->cmsg_type = SCM_RIGHTS; c->cmsg_len = CMSG_LEN(sizeof(int));
  std::memcpy(CMSG_DATA(c), &fd, sizeof(int));
  return sendmsg(sock, &msg, 0) < 0 ? -1 : 0;
}
static int recv_fd(int sock, int* fd, uint64_t* addr, uint64_t* size) {
  struct msghdr msg; std::memset(&msg, 0, sizeof msg);
  uint64_t payload[2] = {0, 0};
  struct iovec io = {payload, sizeof payload};
  char buf[CMSG_SPACE(sizeof(int))]; std::memset(buf, 0, sizeof buf);
  msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = buf; msg.msg_controllen = sizeof buf;
  if (recvmsg(sock, &msg, 0) < 0) return -1;
  struct cmsghdr* c = CMSG_FIRSTHDR(&msg);
  if (!c) return -1;
  std::memcpy(fd, CMSG_DATA(c), sizeof(int));
  *addr = payload[0]; *size = payload[1];
  return 0;
}
int main(int argc, char** argv) {
  if (argc > 2 && !std::strcmp(argv[1], "child")) {
    role = "child";
    const int sock = std::atoi(argv[2]);
    int fd = -1; uint64_t addr = 0, size = 0;
    if (recv_fd(sock, &fd, &addr, &size)) { std::printf("child: recv failed\n"); return 3; }
    CK(hipSetDevice(0));
    hipMemGenericAllocationHandle_t h;
    int ver = 0;
    CK(hipRuntimeGetVersion(&ver));
    std::printf("child: HIP runtime version %d, passing the fd %s\n", ver, getenv("PROBE_PTR") ? "by pointer" : "by value");
    std::fflush(stdout);
    if (getenv("PROBE_PTR")) CK(hipMemImportFromShareableHandle(&h, &fd, hipMemHandleTypePosixFileDescriptor));
    else CK(hipMemImportFromShareableHandle(&h, reinterpret_cast<void*>(static_cast<uintptr_t>(fd)), hipMemHandleTypePosixFileDescriptor));
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, size, 0, reinterpret_cast<void*>(addr), 0));
    std::printf("child: reserved %p (wanted %p)%s\n", va, reinterpret_cast<void*>(addr), va == reinterpret_cast<void*>(addr) ? " SAME" : " DIFFERENT");
    CK(hipMemMap(va, size, 0, h, 0));
    hipMemAccessDesc acc{}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, size, &acc, 1));
    std::vector<uint32_t> host(size / 4);
    CK(hipMemcpy(host.data(), va, size, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < host.size(); i++) bad += host[i] != uint32_t(i * 2654435761u);
    std::printf("child: %zu words read through the imported mapping, %zu wrong\n", host.size(), bad);
    return bad ? 4 : 0;
  }
  int sv[2];
  if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) { std::perror("socketpair"); return 1; }
  const pid_t pid = fork();       // before any HIP call in this process
  if (pid == 0) {
    close(sv[0]);
    char num[16]; std::snprintf(num, sizeof num, "%d", sv[1]);
    execl(argv[0], argv[0], "child", num, static_cast<char*>(nullptr));
    std::perror("execl"); _exit(9);
  }
  close(sv[1]);
  CK(hipSetDevice(0));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  prop.requestedHandleType = hipMemHandleTypePosixFileDescriptor;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t size = ((size_t(8) << 20) + gran - 1) / gran * gran;
  hipMemGenericAllocationHandle_t h;
  CK(hipMemCreate(&h, size, &prop, 0));
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, size, gran, nullptr, 0));
  CK(hipMemMap(va, size, 0, h, 0));
  hipMemAccessDesc acc{}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(va, size, &acc, 1));
  std::vector<uint32_t> host(size / 4);
  for (size_t i = 0; i < host.size(); i++) host[i] = uint32_t(i * 2654435761u);
  CK(hipMemcpy(va, host.data(), size, hipMemcpyHostToDevice));
  int fd = -1;
  CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
  std::printf("parent: granularity %zu, %zu bytes at %p, exported fd %d\n", gran, size, va, fd);
  if (send_fd(sv[0], fd, reinterpret_cast<uint64_t>(va), size)) { std::perror("sendmsg"); return 1; }
  int st = 0;
  waitpid(pid, &st, 0);
  std::printf("parent: child exit %d\n", WIFEXITED(st) ? WEXITSTATUS(st) : -1);
  return WIFEXITED(st) ? WEXITSTATUS(st) : 1;
}
