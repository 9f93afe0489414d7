// SeqNet over TCP: the exchange of the chunked-sequence driver (32 bytes of fingerprints per chunk and round, a ~1.2 MB state blob per mismatching seam between two
// ranks) for ranks that have no RCCL communicator between them -- several ranks rehearsed on one card, or a cluster where only the hosts are connected.  Rank r listens on
// base_port + r; the all-gather is a star through rank 0, the hand-over goes straight from rank r to rank r + 1.  Connections are made on first use and kept.
#include "seq.hpp"
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <thread>

namespace sind {
namespace {

class TcpNet : public SeqNet {
public:
    TcpNet(int rank, int world, std::string host, int base_port) : rank_(rank), world_(world), host_(std::move(host)), base_(base_port), in_star_((size_t)world, -1) {}
    ~TcpNet() override {
        for (int fd : in_star_) if (fd >= 0) ::close(fd);
        for (int fd : {listen_, to_root_, to_next_, from_prev_}) if (fd >= 0) ::close(fd);
    }
    int open_listen() {
        listen_ = ::socket(AF_INET, SOCK_STREAM, 0);
        if (listen_ < 0) return fail("socket");
        int one = 1; (void)::setsockopt(listen_, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a{}; a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_ANY); a.sin_port = htons((uint16_t)(base_ + rank_));
        if (::bind(listen_, (sockaddr*)&a, sizeof(a)) != 0) return fail("bind");
        if (::listen(listen_, world_ + 4) != 0) return fail("listen");
        return 0;
    }
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    const char* error() const override { return err_.c_str(); }

    int allgather(const void* mine, size_t bytes, void* all) override {
        if (world_ == 1) { std::memcpy(all, mine, bytes); return 0; }
        if (rank_ == 0) {
            std::memcpy(all, mine, bytes);
            for (int r = 1; r < world_; r++) { if (in_star_[(size_t)r] < 0 && accept_until(KIND_STAR, r)) return -1; if (recv_all(in_star_[(size_t)r], (uint8_t*)all + (size_t)r * bytes, bytes)) return -1; }
            for (int r = 1; r < world_; r++) if (send_all(in_star_[(size_t)r], all, bytes * (size_t)world_)) return -1;
            return 0;
        }
        if (to_root_ < 0 && connect_to(0, KIND_STAR, &to_root_)) return -1;
        if (send_all(to_root_, mine, bytes)) return -1;
        return recv_all(to_root_, all, bytes * (size_t)world_);
    }
    int sendrecv(const void* send, int to, void* recv, int from, size_t bytes) override {
        if (to >= 0 && to != rank_ + 1) { err_ = "TcpNet::sendrecv: a rank hands over to its successor only"; return -1; }
        if (from >= 0 && from != rank_ - 1) { err_ = "TcpNet::sendrecv: a rank receives from its predecessor only"; return -1; }
        int src = 0; std::thread sender;
        // the send runs beside the receive: a rank in the middle of the chain does both, and a blob is larger than a socket buffer
        if (to >= 0) {
            if (to_next_ < 0 && connect_to(to, KIND_CHAIN, &to_next_)) return -1;
            sender = std::thread([&] { src = send_all(to_next_, send, bytes); });
        }
        int rrc = 0;
        if (from >= 0) { if (from_prev_ < 0) rrc = accept_until(KIND_CHAIN, from); if (!rrc) rrc = recv_all(from_prev_, recv, bytes); }
        if (sender.joinable()) sender.join();
        return (src || rrc) ? -1 : 0;
    }

private:
    enum { KIND_STAR = 1, KIND_CHAIN = 2 };
    int rank_, world_; std::string host_; int base_; std::string err_;
    int listen_ = -1, to_root_ = -1, to_next_ = -1, from_prev_ = -1; std::vector<int> in_star_;
    int fail(const char* what) { err_ = std::string("TcpNet (rank ") + std::to_string(rank_) + "): " + what + ": " + std::strerror(errno); return -1; }
    int send_all(int fd, const void* p, size_t n) {
        const uint8_t* b = (const uint8_t*)p;
        while (n) { const ssize_t k = ::send(fd, b, n, MSG_NOSIGNAL); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return fail("send"); } b += k; n -= (size_t)k; }
        return 0;
    }
    int recv_all(int fd, void* p, size_t n) {
        uint8_t* b = (uint8_t*)p;
        while (n) { const ssize_t k = ::recv(fd, b, n, 0); if (k == 0) { err_ = "TcpNet: peer closed the connection"; return -1; } if (k < 0) { if (errno == EINTR) continue; return fail("recv"); } b += k; n -= (size_t)k; }
        return 0;
    }
    // the peers of a job start at different times: keep trying for two minutes
    int connect_to(int peer, int kind, int* out) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (fd < 0) return fail("socket");
            sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)(base_ + peer));
            if (::inet_pton(AF_INET, host_.c_str(), &a.sin_addr) != 1) { ::close(fd); err_ = "TcpNet: bad host address " + host_; return -1; }
            if (::connect(fd, (sockaddr*)&a, sizeof(a)) == 0) {
                int one = 1; (void)::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
                const int hello[2] = {kind, rank_};
                if (send_all(fd, hello, sizeof(hello))) { ::close(fd); return -1; }
                *out = fd; return 0;
            }
            ::close(fd);
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return fail("connect (peer not up after 120 s)");
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    // accept connections until the one of (kind, peer) is there; others that arrive meanwhile are kept
    int accept_until(int kind, int peer) {
        for (;;) {
            if (kind == KIND_STAR && in_star_[(size_t)peer] >= 0) return 0;
            if (kind == KIND_CHAIN && from_prev_ >= 0) return 0;
            const int fd = ::accept(listen_, nullptr, nullptr);
            if (fd < 0) { if (errno == EINTR) continue; return fail("accept"); }
            int one = 1; (void)::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            int hello[2] = {0, -1};
            if (recv_all(fd, hello, sizeof(hello))) { ::close(fd); return -1; }
            if (hello[0] == KIND_STAR && hello[1] > 0 && hello[1] < world_ && in_star_[(size_t)hello[1]] < 0) in_star_[(size_t)hello[1]] = fd;
            else if (hello[0] == KIND_CHAIN && hello[1] == rank_ - 1 && from_prev_ < 0) from_prev_ = fd;
            else ::close(fd);
        }
    }
};

}  // namespace

SeqNet* seq_net_tcp(int rank, int world, const char* host, int base_port, std::string& err) {
    if (world < 1 || rank < 0 || rank >= world || base_port < 1024 || base_port + world > 65535) { err = "seq_net_tcp: bad rank / world / port"; return nullptr; }
    TcpNet* n = new TcpNet(rank, world, host && *host ? host : "127.0.0.1", base_port);
    if (world > 1 && n->open_listen()) { err = n->error(); delete n; return nullptr; }
    return n;
}

}  // namespace sind
