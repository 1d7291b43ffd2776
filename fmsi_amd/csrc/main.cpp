// `fmsi` command-line entry for the MI355X query path.  Only the sub-commands on the hot path are
// provided (`index`, `query`, `lookup`, `export`; reference dispatcher: src/main.cpp:668-699); everything else of the
// reference's CLI (index construction, set operations, export, ...) is out of scope of this build and
// says so instead of silently doing something different.
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "../../include/fmsi_hip.h"

#define FMSI_AMD_VERSION "0.4.0-mi355x.1"

// fmsi_cli_set_exit_hook: the command's output is complete -- flush, (in the forked child:) close the output and tell the parent, and leave
static int g_done_fd = -1;
static void leave_now(int rc) {
    std::cout.flush();
    std::cerr.flush();
    std::fflush(nullptr);
    if (g_done_fd >= 0) {
        close(STDOUT_FILENO);
        close(STDERR_FILENO);
        const unsigned char code = static_cast<unsigned char>(rc);
        while (write(g_done_fd, &code, 1) < 0 && errno == EINTR) {}
    }
    _exit(rc);
}

static int usage() {
    std::cerr << std::endl;
    std::cerr << "Program: FMSI (MI355X query path) - k-mer membership queries on masked-superstring indexes." << std::endl;
    std::cerr << "Version: " << FMSI_AMD_VERSION << std::endl << std::endl;
    std::cerr << "Usage:   fmsi <command> [options]" << std::endl << std::endl;
    std::cerr << "Command:" << std::endl;
    std::cerr << "    index   - Creates a BWT based index of the given masked superstring (suffix sort on the GPU)." << std::endl;
    std::cerr << "    query   - Queries k-mers against an index (reads the reference's .fmsi.* files)." << std::endl;
    std::cerr << "    lookup  - Return unique hashes of present k-mers." << std::endl;
    std::cerr << "    export  - Export the underlying masked superstring." << std::endl << std::endl;
    std::cerr << "The index files are the reference's .fmsi.* format, in both directions." << std::endl << std::endl;
    return 1;
}

// Does something in this process write its output when the process is finalised -- rocprofv3 / rocprofiler-sdk and rocTX (trace.h's ranges),
// sanitizers, gcov?  A profiler is known by what it puts into the environment (rocprofv3: ROCP_TOOL_LIBRARIES, its tool library preloaded), a
// sanitizer or gcov by how THIS binary was compiled.  (Rounds 2-4 took any LD_PRELOAD and any ASAN_OPTIONS for one -- on a pool whose launcher
// preloads a guard library into every process that made the quick exit dead code.)
static bool needs_full_exit() {
    bool full_exit = false;
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
    full_exit = true;
#elif defined(__has_feature)
#if __has_feature(address_sanitizer) || __has_feature(thread_sanitizer) || __has_feature(memory_sanitizer)
    full_exit = true;
#endif
#endif
    for (const char* v : {"ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB", "GCOV_PREFIX", "FMSI_HIP_CLI_FULL_EXIT"}) {
        const char* e = std::getenv(v);
        if (e && *e) full_exit = true;
    }
    if (const char* pre = std::getenv("LD_PRELOAD"))
        for (const char* tool : {"rocprof", "roctracer", "roctx", "rocsys", "omnitrace", "rocpd"})
            if (std::strstr(pre, tool)) full_exit = true;
    return full_exit;
}

// `index` / `export`: the command has written and closed what it writes; the process leaves without the finalisation of the HIP runtime and
// of this library's statics (round 6: one `fmsi index` in about 600 of the fuzzer's ended with a non-zero status AFTER "Written index" --
// a fault inside that finalisation, nothing of the command's -- and the caller wants none of the 0.1 s it takes anyway)
static int leave_after(int rc) {
    std::cout.flush();
    std::cerr.flush();
    std::fflush(nullptr);
    if (!needs_full_exit()) _exit(rc);
    return rc;
}

int main(int argc, char* argv[]) {
    if (argc < 2) return usage();
    const std::string op = argv[1];
    if (op == "query" || op == "lookup") {
        // The process ends with the command: what it wrote is flushed, and the teardown of the device context, of gigabytes of HBM and
        // of the page-locked batch buffers is left to the operating system (0.1-0.2 s of a 0.9 s run at human scale) -- unless something
        // in the process writes its output when the process is finalised (needs_full_exit).  Then the command tears down what it built (and
        // waits for a dictionary still being built beside the queries) and the ordinary exit runs the atexit handlers and library destructors.
        const bool full_exit = needs_full_exit();
        if (!full_exit) setenv("FMSI_HIP_CLI_NO_TEARDOWN", "1", 0);
        // ... and even left to the operating system, that teardown is 0.11 s + 0.01 s per GB of HBM + what the host side mapped (measured at
        // human scale: 0.14-0.28 s of a 0.7-0.9 s run) in which the caller waits for nothing it wants.  So the command runs in a CHILD, forked
        // here before anything has touched the device; when its output is written it closes stdout and stderr, tells this process its exit
        // code through a pipe and goes on to die at its leisure; this process exits with that code at once.  A child that ends without a word
        // (a signal, the uncaught exception of a missing query file: src/parser.h:22) is waited for and its fate passed on as it is.
        // FMSI_HIP_CLI_NO_FORK=1, and everything that disables the quick exit above, keeps the one process.
        int done_pipe[2] = {-1, -1};
        pid_t child = -1;
        if (!full_exit && !std::getenv("FMSI_HIP_CLI_NO_FORK") && pipe(done_pipe) == 0) {
            const pid_t parent_pid = getpid();
            child = fork();
            if (child < 0) {
                close(done_pipe[0]);
                close(done_pipe[1]);
            } else if (child > 0) {
                close(done_pipe[1]);
                unsigned char code = 0;
                ssize_t got;
                while ((got = read(done_pipe[0], &code, 1)) < 0 && errno == EINTR) {}
                if (got == 1) _exit(code);
                int status = 0;
                while (waitpid(child, &status, 0) < 0 && errno == EINTR) {}
                if (WIFSIGNALED(status)) {
                    signal(WTERMSIG(status), SIG_DFL);
                    raise(WTERMSIG(status));
                    _exit(128 + WTERMSIG(status));
                }
                _exit(WIFEXITED(status) ? WEXITSTATUS(status) : 1);
            } else {
                close(done_pipe[0]);
                g_done_fd = done_pipe[1];
                // the child lives and dies with the process the caller knows: whatever ends the parent early (a `timeout`, a kill of that one
                // PID) ends the child, too -- and when the parent leaves in the ordinary way the child is past its output and only tearing down
                prctl(PR_SET_PDEATHSIG, SIGKILL);
                if (getppid() != parent_pid) _exit(1);   // (the parent was gone before the request stood)
            }
        }
        if (!full_exit) fmsi_cli_set_exit_hook(leave_now);   // (called at the command's last line, before anything is released)
        const int rc = fmsi_cli_query(argc - 1, argv + 1, op == "lookup" ? 1 : 0);   // (comes back on usage errors and under full_exit)
        if (full_exit && !std::getenv("FMSI_HIP_CLI_NO_TEARDOWN")) {
            std::cout.flush();
            std::cerr.flush();
            std::fflush(nullptr);
            std::exit(rc);
        }
        leave_now(rc);
    }
    if (op == "index") return leave_after(fmsi_cli_index(argc - 1, argv + 1));
    if (op == "export") return leave_after(fmsi_cli_export(argc - 1, argv + 1));
    if (op == "-v") {
        std::cout << FMSI_AMD_VERSION << std::endl;
        return 0;
    }
    if (op == "-h") {
        usage();
        return 0;
    }
    if (op == "merge" || op == "compact" || op == "normalize" || op == "clean" ||
        op == "union" || op == "inter" || op == "diff" || op == "symdiff") {
        std::cerr << "ERROR: `fmsi " << op << "` is not part of the MI355X query path; use the reference build for it." << std::endl;
        return 1;
    }
    return usage();
}
