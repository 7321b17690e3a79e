{-# LANGUAGE ForeignFunctionInterface #-}
-- | Binding of libbpgpu.so (include/bpgpu.h) for sdiehl/bulletproofs v1.1.0: the foreign imports, the process-wide
-- engine context and the marshalling of `Fr` / `PA`.
--
-- NOT COMPILED in this repository: the build image has no GHC (DESIGN.md section 3).  This is the source a maintainer
-- adds to the reference package (INTEGRATION.md).  tests/test_shim.py checks every `foreign import` below against
-- include/bpgpu.h (symbol name, number and C type of the arguments, result type).
--
-- The drop-in modules built on it keep the reference's names and types:
--   Bulletproofs.Gpu.Utils              sumExps, addTwoMulP, commit                (Bulletproofs/Utils.hs:71-85)
--   Bulletproofs.Gpu.InnerProductProof  generateProof, verifyProof                 (Bulletproofs/InnerProductProof.hs:1-7)
--   Bulletproofs.Gpu.MultiRangeProof    generateProof(Unsafe), verifyProof         (Bulletproofs/MultiRangeProof.hs:1-8)
--   Bulletproofs.Gpu.RangeProof         generateProof(Unsafe), verifyProof         (Bulletproofs/RangeProof.hs:1-8)
--   Bulletproofs.Gpu.ArithmeticCircuit  generateProof, verifyProof                 (Bulletproofs/ArithmeticCircuit.hs:1-10)
module Bulletproofs.Gpu.FFI
  ( BpCtx, withCtx, maxGenerators, errReferenceFails
  , c_msm, c_add_two_mul, c_point_mul, c_point_add, c_point_neg
  , c_upper_bound_bits, c_log_base2, c_rangeproof_prove, c_rangeproof_verify_batch, c_rangeproof_verify
  , c_ipp_prove, c_ipp_verify, c_ac_proof_points, c_ac_prove, c_ac_verify
  , c_generators_derive, c_generators_load, c_generators_save, c_show_numbers, c_last_error
  , encFr, encFrs, encPt, encPts, decFr, decPt, decPts, withBS, check, leBytes
  ) where

import Protolude
import qualified Data.ByteString as BS
import qualified Data.ByteString.Unsafe as BU
import Data.Bits (shiftL, shiftR, (.|.))
import Foreign
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types
import System.IO.Unsafe (unsafePerformIO)
import Control.Concurrent.MVar (MVar, newMVar, withMVar)

import Data.Curve.Weierstrass.SECP256K1 (PA, Fr)
import Data.Curve.Weierstrass (Point(..))
import Data.Field.Galois (fromP, toP)

data BpCtx

foreign import ccall safe "bp_ctx_create"        c_ctx_create        :: Ptr (Ptr BpCtx) -> CInt -> IO CInt
foreign import ccall safe "bp_last_error"        c_last_error        :: Ptr BpCtx -> IO CString
foreign import ccall safe "bp_generators_derive" c_generators_derive :: Ptr BpCtx -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_generators_load"   c_generators_load   :: Ptr BpCtx -> CString -> IO CInt
foreign import ccall safe "bp_generators_save"   c_generators_save   :: Ptr BpCtx -> CString -> IO CInt
foreign import ccall safe "bp_show_numbers"      c_show_numbers      :: Ptr BpCtx -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_msm"               c_msm               :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_add_two_mul"       c_add_two_mul       :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_point_mul"         c_point_mul         :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_point_add"         c_point_add         :: Ptr BpCtx -> Ptr Word8 -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_point_neg"         c_point_neg         :: Ptr BpCtx -> Ptr Word8 -> CSize -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_upper_bound_bits"  c_upper_bound_bits  :: Ptr Word8 -> CSize -> Ptr Word32 -> IO CInt
foreign import ccall safe "bp_log_base2"         c_log_base2         :: Ptr Word8 -> CSize -> IO CInt
foreign import ccall safe "bp_rangeproof_prove"  c_rangeproof_prove
  :: Ptr BpCtx -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> CInt -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_rangeproof_verify_batch" c_rangeproof_verify_batch
  :: Ptr BpCtx -> Word32 -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_rangeproof_verify" c_rangeproof_verify
  :: Ptr BpCtx -> Ptr Word8 -> CSize -> Ptr Word8 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Word32 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_ipp_prove"         c_ipp_prove
  :: Ptr BpCtx -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_ipp_verify"        c_ipp_verify
  :: Ptr BpCtx -> Word64 -> Ptr Word8 -> Word32 -> Ptr Word8 -> Word32 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_ac_proof_points"   c_ac_proof_points   :: Word32 -> IO CInt
foreign import ccall safe "bp_ac_prove"          c_ac_prove
  :: Ptr BpCtx -> Word32 -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> IO CInt
foreign import ccall safe "bp_ac_verify"         c_ac_verify
  :: Ptr BpCtx -> Word32 -> Word32 -> Word32 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8
  -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> IO CInt

-- | Generator pairs the context derives (`take maxGenerators gs`, `take maxGenerators hs`, Utils.hs:19-25): bounds
-- n * m of range proofs and the gate count of circuits.  8192 is the library's maximum.
maxGenerators :: Int
maxGenerators = 8192

-- | One context per process, like the reference's CAF generator lists (Utils.hs:14-25).  The host application
-- exports GPU_MAX_HW_QUEUES=16 before anything initialises HIP (INTEGRATION.md section 5); the library itself never
-- touches the environment.
--
-- A context is NOT thread-safe (include/bpgpu.h: one stream, grow-only workspaces and a cached verify job that calls
-- replace), and the drop-ins are pure functions that a -threaded program may force from several capabilities at once
-- (`parMap verifyProof`).  Every call on the context therefore goes through `withCtx`, which holds a process-wide
-- lock for the duration of the foreign call AND of the error-message read that may follow it.  The reference is
-- single-threaded, so serialising the calls loses nothing it had; concurrency inside a call (the batch verifier's
-- chunks and hashing threads) is the library's own.
{-# NOINLINE ctxVar #-}
ctxVar :: MVar (Ptr BpCtx)
ctxVar = unsafePerformIO $ do
  c <- alloca $ \pp -> do
    rc <- c_ctx_create pp 0
    when (rc /= 0) $ panic "bp_ctx_create failed"
    c <- peek pp
    rc' <- c_generators_derive c (fromIntegral maxGenerators) nullPtr nullPtr nullPtr
    when (rc' /= 0) $ panic "bp_generators_derive failed"
    pure c
  newMVar c

-- | Run one engine call with the process-wide context, exclusively.  Do not nest.
withCtx :: (Ptr BpCtx -> IO a) -> IO a
withCtx = withMVar ctxVar

-- | BP_ERR_REFERENCE_FAILS (include/bpgpu.h): the reference's own evaluation raises on these arguments.
errReferenceFails :: CInt
errReferenceFails = -8

---------------------------------------------------------------------------------------------------
-- Marshalling: Fr <-> 32 bytes little-endian, PA <-> 64 bytes x || y little-endian, O <-> 64 zero bytes
---------------------------------------------------------------------------------------------------
leBytes :: Int -> Integer -> BS.ByteString
leBytes n v = BS.pack [ fromIntegral (v `shiftR` (8 * i)) | i <- [0 .. n - 1] ]

unLe :: BS.ByteString -> Integer
unLe = BS.foldr (\b acc -> (acc `shiftL` 8) .|. fromIntegral b) 0

encFr :: Fr -> BS.ByteString
encFr = leBytes 32 . fromP

encFrs :: [Fr] -> BS.ByteString
encFrs = BS.concat . fmap encFr

decFr :: BS.ByteString -> Fr
decFr = toP . unLe

encPt :: PA -> BS.ByteString
encPt O       = BS.replicate 64 0
encPt (A x y) = leBytes 32 (fromP x) <> leBytes 32 (fromP y)

encPts :: [PA] -> BS.ByteString
encPts = BS.concat . fmap encPt

decPt :: BS.ByteString -> PA
decPt bs
  | BS.all (== 0) bs = O
  | otherwise        = A (toP (unLe (BS.take 32 bs))) (toP (unLe (BS.drop 32 bs)))

decPts :: Int -> BS.ByteString -> [PA]
decPts n bs = [ decPt (BS.take 64 (BS.drop (64 * i) bs)) | i <- [0 .. n - 1] ]

-- | Pointer to the bytes for the duration of the call; an empty string gives a valid dummy pointer.
withBS :: BS.ByteString -> (Ptr Word8 -> IO a) -> IO a
withBS bs k
  | BS.null bs = allocaBytes 1 k
  | otherwise  = BU.unsafeUseAsCString bs (k . castPtr)

-- | Inside `withCtx`: turn a non-zero status of the call just made on `c` into a `panic` carrying the library's message.
check :: Ptr BpCtx -> Text -> CInt -> IO ()
check c what rc = when (rc /= 0) $ do
  msg <- c_last_error c >>= peekCString
  panic (what <> " failed (" <> show rc <> "): " <> toS msg)
