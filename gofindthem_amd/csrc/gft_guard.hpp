// gft_guard.hpp -- the exception barrier of the C ABI.
//
// The reference returns `error`, it never panics in library code (finder/finder.go:149-158); a C++ exception that crossed
// an extern "C" frame into cgo would be abort() of the Go process.  Every entry point of include/gft.h is therefore a
// function-try-block whose handler is GFT_CATCH(...): whatever the host side throws -- std::bad_alloc from a vector that
// grows with the batch, std::length_error, std::system_error from a thread that could not be started -- comes back as a
// status code with a message, like every other failure.  Host threads (one per device, copy and walk workers) run their
// bodies behind the same translation and are always joined (JoinAll), also when starting one of them failed.
#pragma once
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gft.h"

namespace gft {

// Call from inside a catch (...) handler: the status code of the exception in flight, its message into *err (may be null)
inline int translate_exception(std::string* err) noexcept {
    int code = GFT_E_INTERNAL;
    const char* what = "unknown C++ exception";
    char buf[256];
    try {
        throw;
    } catch (const std::bad_alloc&) {
        code = GFT_E_NOMEM;
        what = "out of host memory";
    } catch (const std::length_error& x) {
        code = GFT_E_NOMEM;                      // a container asked for more than max_size(): the batch is too large
        snprintf(buf, sizeof buf, "host container too large: %s", x.what());
        what = buf;
    } catch (const std::exception& x) {
        snprintf(buf, sizeof buf, "internal error: %s", x.what());
        what = buf;
    } catch (...) {
    }
    if (err) {
        try {
            *err = what;
        } catch (...) {
            // (not even the message fits: the code alone tells)
        }
    }
    return code;
}

// joins every started thread on the way out -- a std::thread that is destroyed while joinable is std::terminate
struct JoinAll {
    std::vector<std::thread>& th;
    explicit JoinAll(std::vector<std::thread>& t) : th(t) {}
    ~JoinAll() {
        for (auto& t : th)
            if (t.joinable()) t.join();
    }
};

}  // namespace gft

// handler of an entry point's function-try-block; ERRP = where the message goes (std::string*, may be null)
#define GFT_CATCH(ERRP) \
    catch (...) { return gft::translate_exception(ERRP); }
// ... for entry points that do not return a status (VALUE = what they return on failure)
#define GFT_CATCH_VALUE(VALUE) \
    catch (...) { (void)gft::translate_exception(nullptr); return VALUE; }
