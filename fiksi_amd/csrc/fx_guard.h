// Nothing leaves an extern "C" function of this library by unwinding (include/fiksi_amd.h "Conventions"). The
// reference's own conventions are what the boundary keeps: System::solve returns (fiksi/src/lib.rs:464-466), numerical
// failure is silent (lm.rs:134-137) — so a host that runs out of memory inside a call gets FX_ERR_NOMEM and an intact
// process, not std::terminate.
//
//   int fx_something(args) try {
//       ...                                   // std::vector, std::thread, new: anything may throw in here
//   }
//   FX_CATCH_CODE                             // -> FX_ERR_NOMEM / FX_ERR_INTERNAL + fx_last_error
//
// Worker threads never let an exception escape either (that would be std::terminate as well): run_workers catches in the
// worker, joins everything, and re-throws the first one on the calling thread, where the entry point's guard turns it into
// a code. A thread that cannot be started is done without: its share runs on the caller.
#pragma once
#include <cstddef>
#include <cstdint>
#include <exception>
#include <new>
#include <thread>
#include <vector>

namespace fx {

// The calling thread's fx_last_error() text lives in a fixed thread-local buffer: reporting "out of memory" allocates nothing.
constexpr size_t LAST_ERROR_LEN = 512;
char* last_error_buffer() noexcept;
void set_last_error(const char* msg) noexcept;
int fail(int code, const char* fmt, ...) noexcept __attribute__((format(printf, 2, 3)));
// inside a catch (...) handler: the status code of the exception in flight (and its text in fx_last_error)
int translate_exception() noexcept;

// work(w) for every w in [0, n): w = 0 on the calling thread, the others on a thread each where one can be had.
template <typename F>
void run_workers(uint32_t n, F&& work) {
    if (n <= 1) {
        if (n) work(0u);
        return;
    }
    std::vector<std::exception_ptr> errs(n);
    std::vector<unsigned char> started(n, 0);
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() {
            for (std::thread& t : th)
                if (t.joinable()) t.join();
        }
    } joiner;
    joiner.th.reserve(n - 1);
    auto guarded = [&](uint32_t w) noexcept {
        try {
            work(w);
        } catch (...) {
            errs[w] = std::current_exception();
        }
    };
    for (uint32_t w = 1; w < n; ++w) {
        try {
            joiner.th.emplace_back(guarded, w);
            started[w] = 1;
        } catch (...) {  // std::system_error (no thread to be had) or std::bad_alloc: the caller takes this share below
        }
    }
    guarded(0u);
    for (uint32_t w = 1; w < n; ++w)
        if (!started[w]) guarded(w);
    for (std::thread& t : joiner.th) t.join();
    for (uint32_t w = 0; w < n; ++w)
        if (errs[w]) std::rethrow_exception(errs[w]);
}

}  // namespace fx

#define FX_CATCH_CODE \
    catch (...) { return ::fx::translate_exception(); }
#define FX_CATCH_VOID \
    catch (...) { (void)::fx::translate_exception(); }
#define FX_CATCH_VALUE(v) \
    catch (...) { (void)::fx::translate_exception(); return (v); }
