// Host-side utilities of the Recommenders assembly.
//   * Tools.printExecutionTime keeps the behaviour of the reference's helper (Recommenders/Tools.cs:6-9; the host
//     calls it once at exit, Program.cs:72): it prints the stopwatch's elapsed whole milliseconds as a TimeSpan.
//   * NativeInfo exposes what the native library reports about itself (not part of the reference's surface).
using System;
using System.Diagnostics;
using System.Runtime.InteropServices;

namespace Recommenders {
    public class Tools {
        public static void printExecutionTime(Stopwatch stopwatch) {
            long elapsedMs = stopwatch.ElapsedMilliseconds;
            Console.WriteLine("Execution time: " + TimeSpan.FromMilliseconds(elapsedMs));
        }
    }

    public static class NativeInfo {
        [DllImport("rwr")] static extern IntPtr rwr_version();
        [DllImport("rwr")] static extern int rwr_device_count();

        /// <summary>Version string of librwr ("x.y.z (gfx950, hip)").</summary>
        public static string Version { get { return Marshal.PtrToStringAnsi(rwr_version()); } }

        /// <summary>Number of usable gfx950 devices; 0 means every compute call will fail (there is no CPU fallback).</summary>
        public static int DeviceCount { get { return rwr_device_count(); } }

        /// <summary>Throws early, with a clear message, when no device is usable.</summary>
        public static void RequireDevice() {
            if (DeviceCount < 1)
                throw new InvalidOperationException("librwr " + Version + ": no usable gfx950 device (no CPU fallback exists)");
        }
    }
}
